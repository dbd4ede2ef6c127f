# which shared unit do the interpenetration kernels wait for?  translation (UTCL2) and atomic counters per kernel, halpe workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
PM="python bench.py --workload pen --steps 1 --warmup 0 --no-cpu --no-alt --no-parity --no-configs3"
rm -rf /tmp/pmc_u /tmp/pmc_a
SFX_PEN_BRANCHES=1 timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY --output-format csv -d /tmp/pmc_u -o p -- $PM > /dev/null 2> gpurun_out/pmc_u.log
SFX_PEN_BRANCHES=1 timeout 600 rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_REQ_sum --output-format csv -d /tmp/pmc_a -o p -- $PM > /dev/null 2> gpurun_out/pmc_a.log
python - <<'P'
import csv, glob, collections
for d in ("/tmp/pmc_u", "/tmp/pmc_a"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"].split("(")[0][:28]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], r["Counter_Name"])
            if r["Counter_Name"] == list(acc[k].keys())[0] and key not in seen: n[k] += 1; seen.add(key)
    print(d)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:16]:
        print("  %-30s launches %6d  " % (k, n[k]) + "  ".join("%s/launch %.0f" % (c, v / max(n[k], 1)) for c, v in acc[k].items()))
P
tail -3 gpurun_out/pmc_u.log gpurun_out/pmc_a.log
