// collide_field.h -- the cone distance field of a triangle and its adjoint (oracle/penetration.py: _cone_geometry, _psi), block reductions
// Part of csrc/collide.hip (included there, in this order: collide_field.h, collide_grid.h, collide_pairs.h, collide_eval.h);
// one translation unit, compiled with -ffp-contract=off.
#pragma once

// ---------------------------------------------------------------------------------------------
// The cone field and its derivatives, written out in reverse mode (round 4; rounds 1-3 pushed forward-mode dual numbers with
// nine tangents through the same formulas: ~10 x the flops of the value, and the pair evaluation was ALU-bound whenever most
// columns of a launch carried the term: p90 155 us).  Notation of oracle/penetration.py.
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(const V3& a, const float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float vdot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// circumscribed circle + unit normal of a triangle (oracle/penetration.py: _cone_geometry); the intermediates the reverse
// sweep needs are kept
struct ConeGeo { V3 a, b, axb, num, oc, o, n; float n2, aa, bb, inv, il, r; };
__device__ __forceinline__ ConeGeo cone_geometry(const V3& p0, const V3& p1, const V3& p2) {
    ConeGeo g;
    g.a = p1 - p0; g.b = p2 - p0;
    g.axb = vcross(g.a, g.b);
    g.n2 = vdot(g.axb, g.axb);
    g.aa = vdot(g.a, g.a); g.bb = vdot(g.b, g.b);
    g.num = vcross(g.axb, g.a) * g.bb + vcross(g.b, g.axb) * g.aa;
    g.inv = 1.f / (g.n2 * 2.f);
    g.oc = g.num * g.inv;
    g.o = p0 + g.oc;
    g.r = sqrtf(vdot(g.oc, g.oc));
    g.il = 1.f / sqrtf(g.n2);
    g.n = g.axb * g.il;
    return g;
}
// adjoint of cone_geometry: (d L / d o, d L / d r, d L / d n) -> d L / d (p0, p1, p2)
__device__ __forceinline__ void cone_geometry_adj(const ConeGeo& g, const V3& go, const float gr, const V3& gn, V3& gp0, V3& gp1, V3& gp2) {
    // r = |oc| (sqrt at 0: zero slope, as the forward-mode version had it); o = p0 + oc
    const V3 goc = go + g.oc * (g.r > 0.f ? gr / g.r : 0.f);
    // n = axb il, il = n2^(-1/2)
    V3 gaxb = gn * g.il;
    float gn2 = vdot(g.axb, gn) * (-0.5f * g.il / g.n2);
    // oc = num inv, inv = 1 / (2 n2)
    const V3 gnum = goc * g.inv;
    gn2 += -vdot(g.num, goc) * g.inv * g.inv * 2.f;
    gaxb = gaxb + g.axb * (2.f * gn2);
    // num = (axb x a) bb + (b x axb) aa
    const V3 u1 = vcross(g.axb, g.a), u2 = vcross(g.b, g.axb);
    const V3 gu1 = gnum * g.bb, gu2 = gnum * g.aa;
    const float gbb = vdot(u1, gnum), gaa = vdot(u2, gnum);
    gaxb = gaxb + vcross(g.a, gu1) + vcross(gu2, g.b);         // u = x x y: dx = y x du, dy = du x x
    V3 ga = vcross(gu1, g.axb) + g.a * (2.f * gaa);
    V3 gb = vcross(g.axb, gu2) + g.b * (2.f * gbb);
    // axb = a x b
    ga = ga + vcross(g.b, gaxb);
    gb = gb + vcross(gaxb, g.a);
    gp1 = ga; gp2 = gb; gp0 = go - ga - gb;
}
// Psi(v)^2 of the cone field (o, r, n) at the point v (oracle/penetration.py: _psi, squared) and its derivatives with respect
// to d = v - o (= d / d v = - d / d o), n and r
__device__ __forceinline__ float cone_penalty(const V3& o, const float r, const V3& n, const V3& v, const float sigma,
                                              const int penalize_outside, V3& gd, V3& gn, float& gr) {
    gd = {0.f, 0.f, 0.f}; gn = {0.f, 0.f, 0.f}; gr = 0.f;
    const V3 d = v - o;
    const float x = vdot(d, n);
    if (!(x < sigma) || (!penalize_outside && x > 0.f)) return 0.f;
    const V3 q = d - n * x;
    const float rho = sqrtf(vdot(q, q));
    const float s = r * (1.f / sigma);
    const float den = r - s * x;
    const float phi = rho / den;
    if (!(phi < 1.f)) return 0.f;
    float ups, dups;
    if (x <= -sigma) { ups = (x * -1.f) + (1.f - sigma); dups = -1.f; }
    else {
        const float c2 = -(1.f - 2.f * sigma) / (4.f * sigma * sigma), c1 = -1.f / (2.f * sigma);
        ups = (x * x) * c2 + x * c1 + ((3.f - 2.f * sigma) / 4.f); dups = 2.f * c2 * x + c1;
    }
    const float w = (1.f - phi) * ups;
    const float psi = w * w;
    // pen = w^4
    const float gw = 4.f * w * psi;
    const float gphi = -ups * gw, gups = (1.f - phi) * gw;
    const float grho = gphi / den, gden = -gphi * phi / den;           // phi = rho / den
    gr = gden * (1.f - x * (1.f / sigma));                              // den = r - (r / sigma) x
    float gx = gups * dups - gden * s;
    const V3 gq = q * (rho > 0.f ? grho / rho : 0.f);                   // rho = |q|
    gx -= vdot(n, gq);                                                  // q = d - n x
    gd = gq + n * gx;                                                   // x = d . n
    gn = gq * (-x) + d * gx;
    return psi * psi;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_min(float v, float* red) {
    v = -wave_max_dpp(-v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < PEN_T / 64; ++i) r = fminf(r, red[i]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) { return -block_min(-v, red); }
__device__ __forceinline__ float block_sum_fixed(float v, float* red) {
    v = wave_sum_dpp(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < PEN_T / 64; ++i) r += red[i];
    __syncthreads();
    return r;
}

// exclusive prefix sum over the PEN_T lanes of the block (fixed order); *total = sum of all
__device__ __forceinline__ int block_excl_scan(const int v, int* wsum /* [PEN_T / 64] */, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = wave_incl_scan_dpp(v);
    __syncthreads();
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < PEN_T / 64; ++i) { const int x = wsum[i]; if (i < wv) base += x; tot += x; }
    *total = tot;
    return base + inc - v;
}

