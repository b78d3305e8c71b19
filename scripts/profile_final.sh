#!/bin/bash
# profiles for the round: kernel trace of the default bench command, PMC HBM traffic of the dominant kernel, MLP GEMM-phase kernel
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
R=${1:-r02}
RP="timeout -k 5 600 rocprofv3"     # every profiler pass under its own limit: a counter set the tool rejects must not hold the box
rm -rf gpurun_out/pf; mkdir -p gpurun_out/pf/kt gpurun_out/pf/fetch gpurun_out/pf/write gpurun_out/pf/mlp gpurun_out/pf/tr gpurun_out/out
python3 bench.py > gpurun_out/out/${R}_bench_default.json 2> gpurun_out/out/${R}_bench_default.err
$RP --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pf/kt/log 2>&1
grep '^{' gpurun_out/pf/kt/log | tail -1 > gpurun_out/out/${R}_bench_under_rocprof.json
cp gpurun_out/pf/kt/kt_kernel_stats.csv gpurun_out/out/${R}_kernel_stats.csv
$RP --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf/fetch -o f -- python3 bench.py --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/pf/fetch/log 2>&1
$RP --pmc WRITE_SIZE --output-format csv -d gpurun_out/pf/write -o w -- python3 bench.py --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/pf/write/log 2>&1
$RP --kernel-trace --stats --output-format csv -d gpurun_out/pf/mlp -o m -- python3 scripts/dev_mlp_bench.py > gpurun_out/pf/mlp/log 2>&1
cp gpurun_out/pf/mlp/m_kernel_stats.csv gpurun_out/out/${R}_mlp_kernel_stats.csv
grep -E '^(mlp_mfma|valu)' gpurun_out/pf/mlp/log > gpurun_out/out/${R}_mlp_bench.txt
$RP --kernel-trace --stats --output-format csv -d gpurun_out/pf/tr -o t -- python3 scripts/train_step_bench.py --steps 3 > gpurun_out/pf/tr.log 2>&1
cp gpurun_out/pf/tr/t_kernel_stats.csv gpurun_out/out/${R}_train_step_kernel_stats.csv
grep '^stage-1' gpurun_out/pf/tr.log > gpurun_out/out/${R}_train_step.txt
python3 - "$R" <<'PY'
import csv, glob, collections, json, sys
R = sys.argv[1]
out = {}
for name, d in (('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')):
    fs = glob.glob('gpurun_out/pf/%s/**/*counter_collection.csv' % d, recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    if fs:
        for r in csv.DictReader(open(fs[0])):
            if r.get('Counter_Name') != name: continue
            k = r['Kernel_Name'].split('(')[0]
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    out[name] = {k: {'sum_KB': v[0], 'launches': v[1], 'avg_KB': v[0] / v[1]} for k, v in agg.items()}
json.dump(out, open('gpurun_out/out/%s_pmc_raw.json' % R, 'w'), indent=1)
def pick(d, key):      # all instantiations of the kernel (ray queue / pixel-pair queue) together: launch-weighted average
    tot, n = 0.0, 0
    for k, v in d.items():
        if key in k: tot += v['sum_KB']; n += v['launches']
    return {'avg_KB': tot / n if n else 0.0, 'launches': n}
f = pick(out['FETCH_SIZE'], 'k_trace_any4q<false'); w = pick(out['WRITE_SIZE'], 'k_trace_any4q<false')
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline",
           "kernel": "mr::k_trace_any4q<false, *> (all shadow-ray launches of the frame: ray queues and the spatial pass's pixel-pair queue)", "launches_sampled": f['launches'], "FETCH_SIZE_KB_avg": f['avg_KB'], "WRITE_SIZE_KB_avg": w['avg_KB'],
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of 16-B/lane loads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE as reported; the guide calibrates the factor on streaming reads, so for the traversal's dwordx4 gathers it is an extrapolation",
           "k_trace_any_hbm_bytes_per_launch": round((2 * f['avg_KB'] + w['avg_KB']) * 1024)}, open('gpurun_out/out/pmc_traffic.json', 'w'), indent=1)
print(open('gpurun_out/out/pmc_traffic.json').read())
PY
# round 2: the shadow-ray kernel alone (frame-like ray set): SQ / TA / TCP / TD counters per launch, per-phase wave cycles, instruction issue costs
bash scripts/pmc_any.sh any 1600 7 3 0 > gpurun_out/out/${R}_pmc_any4q.txt 2>&1
cp gpurun_out/pmc_any/summary.json gpurun_out/out/${R}_pmc_any4q.json 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && timeout -k 3 120 /tmp/valu_rates > gpurun_out/out/${R}_valu_rates.txt 2>&1
rm -rf gpurun_out/pf
cat gpurun_out/out/${R}_bench_default.json | cut -c1-1800
head -12 gpurun_out/out/${R}_kernel_stats.csv | cut -c1-160
cat gpurun_out/out/${R}_mlp_bench.txt
