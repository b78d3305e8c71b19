#!/bin/bash
# Round-4 profile collection (GPU box): scripts/profile_final.sh r04 (default bench line, kernel trace of the same command, FETCH / WRITE traffic of the shadow-ray kernel,
# MLP GEMM phase, training step, SQ / TA / TCP counters of the shadow-ray kernel alone on both meshes) + the MLP's matrix-pipe counters + the derived figures bench.py's
# roofline quotes — profiles/pmc_any4q_summary.json and profiles/pmc_traffic.json, each stamped with csrc_sha (bench.csrc_sha(): the hash of the kernel sources
# they were measured on; bench.py reports them only while it matches). Every rocprofv3 pass: counters OR kernel trace, never both; the program itself after `--`.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
SHA=$(python3 -c "import bench; print(bench.csrc_sha())")
bash scripts/profile_final.sh r04 > gpurun_out/profile_r04.log 2>&1
bash scripts/pmc_mlp.sh > gpurun_out/out/r04_pmc_mlp.txt 2>&1
MIRRES_MESH=clustered bash scripts/pmc_any.sh any_clustered 1600 7 3 0 > gpurun_out/out/r04_pmc_any4q_clustered.txt 2>&1
cp gpurun_out/pmc_any_clustered/summary.json gpurun_out/out/r04_pmc_any4q_clustered.json 2>/dev/null
python3 - "$SHA" <<'PY'
import json, re, sys
sha = sys.argv[1]
def derive(path):
    d = json.load(open(path))
    k = [x for x in d if 'any4q' in x][0]; c = d[k]
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0                      # GRBM_GUI_ACTIVE sums the eight XCDs
    return dict(kernel=k, kernel_cycles=cyc,
                valu_busy=round(4.0 * c['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc), 4),      # SQ_ACTIVE_INST_* count quad-cycles; 1024 SIMDs
                lane_util=round(c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU']), 4),
                l1_hit=round(1.0 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum'], 4),
                wait_any_of_wave_cycles=round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4),
                ta_busy=round(c['TA_TA_BUSY_sum'] / (256.0 * cyc), 4), td_busy=round(c['TD_TD_BUSY_sum'] / (256.0 * cyc), 4),
                valu_insts_per_launch=c.get('SQ_INSTS_VALU'))
out = {"csrc_sha": sha}
try:
    out.update(derive('gpurun_out/out/r04_pmc_any4q.json'))
    out["source"] = "scripts/pmc_any.sh: the shadow-ray kernel alone on 6.9 M frame-like rays (icosphere), separate rocprofv3 --pmc passes, per-launch averages"
except Exception as e:
    out['any4q_error'] = repr(e)
try:
    out["clustered"] = derive('gpurun_out/out/r04_pmc_any4q_clustered.json')
except Exception as e:
    out['clustered_error'] = repr(e)
try:
    t = open('gpurun_out/out/r04_pmc_mlp.txt').read()
    m = re.search(r"k_mlp_mfma<0, 2> (\{.*?\}) launches", t)
    c = eval(m.group(1))
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0
    out.update(mlp_source="scripts/pmc_mlp.sh: GEMM phase of the material MLP (2.56 M points), millions per launch", mlp_kernel_cycles_M=cyc,
               mlp_mfma_busy=round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc, 4), mlp_mfma_instructions_M=c['SQ_INSTS_MFMA'])
except Exception as e:
    out['mlp_error'] = repr(e)
json.dump(out, open('gpurun_out/out/pmc_any4q_summary.json', 'w'), indent=1)
try:
    tr = json.load(open('gpurun_out/out/pmc_traffic.json')); tr["csrc_sha"] = sha
    json.dump(tr, open('gpurun_out/out/pmc_traffic.json', 'w'), indent=1)
except Exception as e:
    print("traffic:", e)
print(json.dumps(out, indent=1))
PY
tail -5 gpurun_out/profile_r04.log
