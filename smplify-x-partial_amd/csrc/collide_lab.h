// collide_lab.h -- LAB BUILD ONLY (-DSFX_LAB, libsfx_lab.so; include/sfx_lab.h): what the measurements of rounds 3-5 need and the
// product does not -- the second and third form of the interpenetration step (k_pen_narrow as the per-column fast path with the
// hand-over to the general kernels), the phase clocks of the kernels.  Included by collide.hip behind `struct sfx_pen`; every form
// produces the bits of the product's (tests/test_gpu_topology.py::test_the_forms_of_the_term_give_the_same_bits, run by
// tools/run_gpu_suite.sh on the lab build).  LAB_NOTES.md has the measurements that decided against these forms.
#pragma once

// which form of the term new handles take (sfx_debug_pen_form): 0 = the product's -- the general kernels on every column, every
// step dealt flat over the chip (fastest on the halpe cfg's fit: 332 frames/s in round 5's A/B); 1 = grid build and pair tests over
// the chip, then one workgroup per column behind the pairs (k_pen_narrow) + the general kernels on the columns it hands over (296
// frames/s: a round lasts as long as its most crowded column's workgroup, and the fits always carry a few collapsed meshes); 2 = form
// 1 with every column handed over (exercises the hand-over on any mesh).  (Round 5's form 3 -- one workgroup per column behind the
// triangle boxes, k_pen_frame, 145 frames/s -- was deleted in round 6: commit e8a08e9 is the last tree that holds it.)  Same bits in
// every form (tests/test_gpu_topology.py); LAB_NOTES.md has the measurements.
static int g_pen_form = [] { const char* e = getenv("SFX_PEN_FORM"); return e ? atoi(e) : 0; }();
extern "C" int sfx_debug_pen_form(int32_t form) {
    const int prev = g_pen_form;
    if (form >= 0 && form <= 2) g_pen_form = form;
    return prev;
}


// debug: wall-clock ticks (100 MHz) k_pen_narrow's workgroups spent in their phases since sfx_pen_work_reset, summed over the column
// evaluations that went through them: [0] until the pairs are read (entry), [1] D pair list, [2] E pair evaluation, [3] F triangle
// sums, [4] G vertices and loss, [5] number of such evaluations, [6] their ordered pairs, [7] unused
extern "C" int sfx_debug_pen_phase_ticks(int64_t* out /* [8] */) {
    if (!out) { sfx_set_error("null argument"); return -1; }
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!g_pen_work) return 0;
    if (hipDeviceSynchronize() != hipSuccess) { sfx_set_error("device error"); return -4; }
    unsigned long long h[8];
    if (hipMemcpy(h, g_pen_work + 8, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { sfx_set_error("device error"); return -4; }
    for (int i = 0; i < 8; ++i) out[i] = (int64_t)h[i];
    return 0;
}

// debug: wall-clock ticks (100 MHz) at the end of k_pen_pairs' steps for the first B frames: [B][10] = triangle boxes,
// frame box, part boxes, part culling, grid histogram, scan, scatter ([7..9] unused: those steps are kernels of their own); [10] = grid entries
extern "C" int sfx_pen_phase_clocks(sfx_pen* h, int32_t B, int32_t* out) {
    if (!h || !out || B < 1 || B > h->Bmax) return -1;
    std::vector<int> st((size_t)B * PEN_STATS);
    hipDeviceSynchronize();
    hipMemcpy(st.data(), h->P.stats, st.size() * sizeof(int), hipMemcpyDeviceToHost);
    for (int i = 0; i < B; ++i) for (int k = 0; k < 11; ++k) out[i * 11 + k] = st[(size_t)i * PEN_STATS + 4 + k];
#ifdef PEN_COUNT
    {   // per frame: grid entries, candidates by the test they die on, wavefront steps of the walk -- and how unevenly the frames carry them
        long tot[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < B; ++i) {
            const int* r = &st[(size_t)i * PEN_STATS];
            const long v[6] = {r[14], r[16], r[17], r[18], r[19], r[20]};
            for (int q = 0; q < 6; ++q) { tot[q] += v[q]; mx[q] = std::max(mx[q], v[q]); }
        }
        long ph[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < B; ++i) for (int q = 0; q < 6; ++q) ph[q] += st[(size_t)i * PEN_STATS + 24 + q];
        fprintf(stderr, "[pen count] k_pen_g3 mean shader cycles per phase: init %ld, part masks %ld, histogram %ld, scan %ld, scatter %ld, copy %ld\n",
                ph[0] / B, ph[1] / B, ph[2] / B, ph[3] / B, ph[4] / B, ph[5] / B);
        fprintf(stderr, "[pen count] %d frames, mean / max per frame: entries %ld / %ld; walked %ld / %ld, same cell %ld / %ld, part mask passed %ld / %ld, "
                "boxes overlap %ld / %ld; wavefront steps %ld / %ld\n", B, tot[0] / B, mx[0], tot[1] / B, mx[1], tot[2] / B, mx[2], tot[3] / B, mx[3],
                tot[4] / B, mx[4], tot[5] / B, mx[5]);
    }
#endif
    return 0;
}

#define PEN_HEAVY_ROWS 8        // grid rows of the general kernels when they work on the handed-over columns (they loop over the list)
// forms 1 / 2: grid build and pair tests over the chip, the pairs into one list per column, ONE workgroup per column behind them
// (k_pen_narrow); form 2 hands every column over to the general kernels (exercises the hand-over on any mesh)
static int pen_eval_cols_narrow(sfx_pen* h, const PenDev& P0, const PenDev& Pl, int32_t B, const float* verts_dev, float sigma, int32_t penalize_outside,
                                float* loss_dev, float* dverts_dev, const int* want_dev, const PenAdjPrep& ap, size_t list_lds, size_t rank_lds,
                                int rank_rows, int cap_pad, hipStream_t s) {
    static bool frame_attr = false;
    const size_t narrow_lds = (size_t)(2 * PEN_FP + 2 * P0.hasp_words + (P0.V + 31) / 32 + P0.V) * sizeof(int);
    if (!frame_attr) {
        if (hipFuncSetAttribute((const void*)k_pen_narrow<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_pen_narrow<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
            sfx_set_error("cannot reserve LDS for k_pen_narrow"); return -2; }
        frame_attr = true;
    }
    const PenSel all{want_dev, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_pen_g1, dim3(PEN_GW, (B + 7) & ~7), dim3(PEN_T), 0, s, P0, verts_dev, want_dev, dverts_dev, ap.adj_G, ap.Vpad, B, (float*)nullptr);
    hipLaunchKernelGGL(k_pen_g2, dim3(PEN_GW, (B + 7) & ~7), dim3(PEN_T), 0, s, P0, want_dev, B);
    hipLaunchKernelGGL(k_pen_g3, dim3(B), dim3(PEN_T), (size_t)(PEN_GRID_INTS + PEN_CELLS) * sizeof(int), s, P0, want_dev);
    hipLaunchKernelGGL(k_pen_walk, dim3(PEN_WALK_BLOCKS, B), dim3(256), 0, s, P0, all, 1, 0);
    hipLaunchKernelGGL(k_pen_walk2, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, B, all, 1);
    if (P0.p2p) hipLaunchKernelGGL(k_pen_narrow<true>, dim3(B), dim3(PEN_T), narrow_lds, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, want_dev, ap, h->form == 2 ? 1 : 0);
    else hipLaunchKernelGGL(k_pen_narrow<false>, dim3(B), dim3(PEN_T), narrow_lds, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, want_dev, ap, h->form == 2 ? 1 : 0);
    // the general kernels on the columns handed over (usually none: each of these seven launches then ends after one load)
    const PenSel hv{nullptr, P0.hlist, P0.nheavy, P0.heavy};
    const int HY = std::min(B, PEN_HEAVY_ROWS);
    hipLaunchKernelGGL(k_pen_walk, dim3(PEN_WALK_BLOCKS, HY), dim3(256), 0, s, P0, hv, 0, 0);
    hipLaunchKernelGGL(k_pen_walk2, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, B, hv, 0);
    hipLaunchKernelGGL(k_pen_list, dim3(HY), dim3(PEN_T), list_lds, s, Pl, hv);
    hipLaunchKernelGGL(k_pen_rank, dim3(rank_rows, HY), dim3(256), rank_lds, s, P0, hv, cap_pad, 0);
    if (P0.p2p) hipLaunchKernelGGL(k_pen_eval<true>, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, verts_dev, sigma, penalize_outside, B, 1, hv);
    else hipLaunchKernelGGL(k_pen_eval<false>, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, verts_dev, sigma, penalize_outside, B, 1, hv);
    hipLaunchKernelGGL(k_pen_facesum, dim3(PEN_EVAL_BLOCKS, HY), dim3(256), 0, s, P0, hv);
    hipLaunchKernelGGL(k_pen_gather, dim3((std::max(P0.V, 1) + 255) / 256, HY), dim3(256), (size_t)P0.hasp_words * sizeof(unsigned), s,
                       P0, dverts_dev, loss_dev, hv, ap);
    if (hipGetLastError() != hipSuccess) { sfx_set_error("penetration kernels failed to launch"); return -4; }
    return 0;
}
