#!/bin/bash
# Round-6 profile collection (GPU box), ONE run on the frozen kernel sources. Order matters: the counter snapshots are collected FIRST and copied into this box's profiles/
# (each stamped with the hash of the source GROUP it depends on — bench.csrc_sha(group): a change in raster.hip no longer stales the traversal counters), so that the bench
# lines produced afterwards in the same call quote them as fresh; everything lands in gpurun_out/out/ and is copied into profiles/ by the builder.
# Every rocprofv3 pass: counters OR kernel trace, never both; the program itself after `--`; each pass under its own timeout.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
R=r06
O=gpurun_out/out; mkdir -p $O
python3 -c "import bench; print('csrc_sha', bench.csrc_sha(), {g: bench.csrc_sha(g) for g in bench.SHA_GROUPS})" | tee $O/${R}_csrc_sha.txt
RP="timeout -k 5 600 rocprofv3"

# ---- 1. counters: the shadow-ray kernel alone (both meshes), every kernel of one serialised frame (both meshes), MLP GEMM phase, the training step's backward kernels
# The counter passes render SHORT frames (8 spp) of the metric's scene: the hierarchy is fixed to the one the metric's 512-spp frame traverses (round 6: update_mesh builds
# the extended-Morton tree and a long frame adds the SAH top; a short one would not)
export MIRRES_PRIVATE_TREE=2
bash scripts/pmc_any.sh any 1600 7 3 0 > $O/${R}_pmc_any4q.txt 2>&1; cp gpurun_out/pmc_any/summary.json $O/${R}_pmc_any4q.json 2>/dev/null
MIRRES_MESH=clustered bash scripts/pmc_any.sh any_clustered 1600 7 3 0 > $O/${R}_pmc_any4q_clustered.txt 2>&1; cp gpurun_out/pmc_any_clustered/summary.json $O/${R}_pmc_any4q_clustered.json 2>/dev/null
bash scripts/pmc_chain.sh chain_ico icosphere 8 > $O/${R}_pmc_chain_icosphere.txt 2>&1; cp gpurun_out/pmc_chain_ico/summary.json $O/${R}_pmc_chain_icosphere.json 2>/dev/null
bash scripts/pmc_chain.sh chain_clu clustered 8 > $O/${R}_pmc_chain_clustered.txt 2>&1; cp gpurun_out/pmc_chain_clu/summary.json $O/${R}_pmc_chain_clustered.json 2>/dev/null
bash scripts/pmc_mlp.sh > $O/${R}_pmc_mlp.txt 2>&1
unset MIRRES_PRIVATE_TREE      # the training step and every bench line below: the default (two-step) build
bash scripts/pmc_train.sh train > $O/${R}_pmc_train.txt 2>&1; cp gpurun_out/pmc_train/summary.json $O/${R}_pmc_train.json 2>/dev/null
python3 scripts/dev_kernel_regs.py > $O/${R}_kernel_regs.txt 2>&1        # code-object metadata: what occupancy is computed from

# ---- 2. the derived figures bench.py quotes
python3 - "$R" <<'PY'
import json, re, sys, subprocess
sys.path.insert(0, '.')
import bench
R = sys.argv[1]
O = 'gpurun_out/out/'
sha = {g: bench.csrc_sha(g) for g in bench.SHA_GROUPS}
# registers from the code object (the kernel trace's VGPR_Count is half the allocation): mangled-name fragment -> total VGPR + AGPR
regs = {}
for l in open(O + R + '_kernel_regs.txt'):
    m = re.match(r"\S+\s+(\S+)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)", l)
    if m: regs[m.group(1)] = int(m.group(2))
def reg_of(frag):
    c = [v for k, v in regs.items() if frag in k]
    return max(c) if c else None
def derive_any(path):
    d = json.load(open(path))
    k = [x for x in d if 'any4q' in x][0]; c = d[k]
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0                      # GRBM_GUI_ACTIVE sums the eight XCDs
    return dict(kernel=k, kernel_cycles=cyc,
                valu_busy=round(min(1.0, 4.0 * c['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc)), 4),      # SQ_ACTIVE_INST_* count quad-cycles; 1024 SIMDs
                lane_util=round(c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU']), 4),
                l1_hit=round(1.0 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum'], 4),
                wait_any_of_wave_cycles=round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4),
                ta_busy=round(c['TA_TA_BUSY_sum'] / (256.0 * cyc), 4), td_busy=round(c['TD_TD_BUSY_sum'] / (256.0 * cyc), 4),
                valu_insts_per_launch=c.get('SQ_INSTS_VALU'), salu_insts_per_launch=c.get('SQ_INSTS_SALU'))
out = {"csrc_sha": sha["traversal"], "csrc_sha_group": "traversal"}
try:
    out.update(derive_any(O + R + '_pmc_any4q.json'))
    out["source"] = "scripts/pmc_any.sh: the shadow-ray kernel alone on 6.9 M frame-like rays (icosphere), separate rocprofv3 --pmc passes, per-launch averages"
except Exception as e:
    out['any4q_error'] = repr(e)
try:
    out["clustered"] = derive_any(O + R + '_pmc_any4q_clustered.json')
except Exception as e:
    out['clustered_error'] = repr(e)
try:
    t = open(O + R + '_pmc_mlp.txt').read()
    m = re.search(r"k_mlp_mfma<0, 2> (\{.*?\}) launches", t)
    c = eval(m.group(1))
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0
    out.update(mlp_source="scripts/pmc_mlp.sh: GEMM phase of the material MLP (2.56 M points), millions per launch", mlp_kernel_cycles_M=cyc,
               mlp_mfma_busy=round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc, 4), mlp_mfma_instructions_M=c['SQ_INSTS_MFMA'])
except Exception as e:
    out['mlp_error'] = repr(e)
json.dump(out, open(O + 'pmc_any4q_summary.json', 'w'), indent=1)

chain = {"csrc_sha": sha["chain"], "csrc_sha_group": "chain",
         "source": "scripts/pmc_chain.sh: MIRRES_STREAMS=1 rocprofv3 --pmc <set> -- python3 bench.py --mesh <mesh> --spp 8 --steps 1 --warmup 0 (separate passes), per-launch averages; "
                   "valu_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMD x kernel cycles), lane_util = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), hbm = (2 FETCH_SIZE + WRITE_SIZE) x 1024 B; "
                   "vgpr = the code object's .vgpr_count (arch VGPRs + AGPRs: what the allocation granule and the waves per SIMD follow)"}
traffic = {"csrc_sha": sha["chain"], "csrc_sha_group": "chain", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of scripts/pmc_chain.sh) over one serialised 8-spp frame per mesh",
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of 16-B/lane loads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE as reported; the guide calibrates the factor on streaming reads, so for the traversal's dwordx4 gathers it is an extrapolation"}
want = {"k_spatial_resolve<5, true>": ("spatial_resolve", "k_spatial_resolveILi5ELb1E"), "k_spatial_gen<5, true>": ("spatial_gen", "k_spatial_genILi5ELb1E"), "k_initial_gen<true>": ("initial_gen", "k_initial_genILb1E"),
        "k_trace_closest4<false>": ("closest4", "k_trace_closest4ILb0E"), "k_bounce_gen": ("bounce_gen", "k_bounce_gen"), "k_new_dir_gen": ("new_dir_gen", "k_new_dir_gen"), "k_mlp_mfma<1, 2>": ("mlp_fused", "k_mlp_mfmaILi1ELi2E")}
for mesh in ("icosphere", "clustered"):
    try:
        d = json.load(open(O + R + '_pmc_chain_%s.json' % mesh))['kernels']
    except Exception as e:
        chain[mesh + '_error'] = repr(e); continue
    rec = {}
    for k, (short, frag) in want.items():
        x = d.get(k)
        if not x: continue
        us = x.get('kernel_cycles', 0) / 2400.0
        rec[short] = {"kernel": k, "launch_us_at_2.4GHz": round(us, 1), "valu_busy": round(min(1.0, x.get('valu_busy', 0)), 4), "lane_util": round(x.get('lane_util', 0), 4),
                      "valu_useful": round(min(1.0, x.get('valu_busy', 0)) * x.get('lane_util', 0), 4), "wait_any_of_wave_cycles": round(x.get('wait_any_of_wave_cycles', 0), 4),
                      "l1_hit": round(x.get('l1_hit', 0), 4), "valu_insts_per_wave": round(x.get('valu_insts_per_wave', 0)), "waves": round(x.get('SQ_WAVES', 0)),
                      "vgpr": reg_of(frag), "hbm_bytes_per_launch": round(x.get('hbm_bytes_per_launch_fetch2x_plus_write', 0)),
                      "hbm_frac_of_8TBps": round(x.get('hbm_bytes_per_launch_fetch2x_plus_write', 0) / max(us, 1e-9) / 1e6 / 8.0, 4)}
    per_frame = ('k_eaw', 'k_matnet_fwd', 'k_trace_persist', 'k_sah', 'k_refit', 'k_rs_', 'k_env_', 'k_neighbor', 'k_morton', 'k_elements', 'k_hierarchy', 'k_pack', 'k_emc', 'k_flip', 'k_own_occ',
                 'k_init_extent', 'k_gbuf', 'k_finish', 'k_average', 'k_composite', 'k_normal_ao', 'k_light_tables')
    tot = sum(x['SQ_INSTS_VALU'] * x['_launches'] for k, x in d.items() if 'SQ_INSTS_VALU' in x and not k.startswith(per_frame))
    rec['_sample'] = {"valu_wave_insts_per_sample": round(tot / 8.0), "spp_of_the_profiled_frame": 8,
                      "note": "sum over the per-sample / per-batch kernels of SQ_INSTS_VALU x launches / 8; the VALU-issue roof is 1024 SIMDs x clock / 4 cycles per wave64 instruction"}
    chain[mesh] = rec
    anyk = [k for k in d if k.startswith('k_trace_any4q<false')]
    tot = sum(d[k].get('hbm_bytes_per_launch_fetch2x_plus_write', 0) * d[k]['_launches'] for k in anyk); n = sum(d[k]['_launches'] for k in anyk)
    vi = sum(d[k].get('SQ_INSTS_VALU', 0) * d[k]['_launches'] for k in anyk)
    t = {"kernel": "mr::k_trace_any4q<false, *> (all shadow-ray launches of the frame: ray queues and the spatial pass's pixel-pair queue)", "launches_sampled": n,
         "k_trace_any_hbm_bytes_per_launch": round(tot / max(1, n)), "k_trace_any_valu_insts_per_launch": round(vi / max(1, n))}
    if mesh == "icosphere": traffic.update(t)
    else: traffic[mesh] = t
json.dump(chain, open(O + 'pmc_chain_summary.json', 'w'), indent=1)
json.dump(traffic, open(O + 'pmc_traffic.json', 'w'), indent=1)

# the training step's backward kernels (scripts/pmc_train.sh)
train = {"csrc_sha": sha["train"], "csrc_sha_group": "train", "source": "scripts/pmc_train.sh: rocprofv3 --pmc <set> -- python3 scripts/train_step_bench.py --steps 2 (separate passes), per-launch averages of the backward kernels; "
         "atomic requests = TCC_ATOMIC_sum (= TCC_EA0_ATOMIC_sum: every one leaves the L2); roof = 21 G requests/s (profiles/r06_atomic_rate.txt)"}
try:
    d = json.load(open(O + R + '_pmc_train.json'))
    for k, c in d.items():
        short = k.split('<')[0]
        train[short] = {"kernel": k, "launch_us": round(c.get('launch_us_at_2.4GHz', 0), 1), "atomic_requests_per_launch": round(c.get('TCC_ATOMIC_sum', 0)),
                        "valu_busy": round(c.get('valu_busy', 0), 4), "lane_util": round(c.get('lane_util', 0), 4), "wait_any_of_wave_cycles": round(c.get('wait_any_of_wave_cycles', 0), 4),
                        "wait_inst_of_wave_cycles": round(c.get('SQ_WAIT_INST_ANY', 0) / max(1.0, c.get('SQ_WAVE_CYCLES', 1.0)), 4),
                        "hbm_bytes_per_launch": round(c.get('hbm_bytes_per_launch_fetch2x_plus_write', 0)), "vgpr": reg_of(short), "launches": c.get('_launches')}
except Exception as e:
    train['error'] = repr(e)
json.dump(train, open(O + 'pmc_train_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:1500]); print(json.dumps(traffic, indent=1)[:900]); print(json.dumps(train, indent=1)[:1500])
PY
cp $O/pmc_any4q_summary.json $O/pmc_chain_summary.json $O/pmc_traffic.json $O/pmc_train_summary.json profiles/      # this box's copy: the bench lines below quote them as fresh

# ---- 3. bench lines (default, driver-style, lego-like as the headline), kernel traces of both meshes, training step, MLP GEMM phase
timeout -k 5 600 python3 bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err
timeout -k 5 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_driver_style.json 2> $O/${R}_bench_driver_style.err
timeout -k 5 600 python3 bench.py --mesh clustered --no-extras > $O/${R}_bench_clustered.json 2> $O/${R}_bench_clustered.err
mkdir -p gpurun_out/pf
for mesh in icosphere clustered; do
  rm -rf gpurun_out/pf/kt; $RP --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o kt -- python3 bench.py --mesh $mesh --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pf/log_$mesh 2>&1
  grep '^{' gpurun_out/pf/log_$mesh | tail -1 > $O/${R}_bench_under_rocprof_$mesh.json
  find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_kernel_stats_$mesh.csv
done
rm -rf gpurun_out/pf/kt; $RP --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o m -- python3 scripts/dev_mlp_bench.py > gpurun_out/pf/log_mlp 2>&1
find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_mlp_kernel_stats.csv; grep -E '^(mlp_mfma|valu)' gpurun_out/pf/log_mlp > $O/${R}_mlp_bench.txt
rm -rf gpurun_out/pf/kt; $RP --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o t -- python3 scripts/train_step_bench.py --steps 3 > gpurun_out/pf/log_tr 2>&1
find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_train_step_kernel_stats.csv; grep '^stage-1' gpurun_out/pf/log_tr > $O/${R}_train_step.txt
MIRRES_PRIVATE_TREE=2 timeout -k 5 300 python3 scripts/dev_leaf_branch.py 4 2>&1 | grep -v amdgpu.ids > $O/${R}_leaf_branch.txt
rm -rf gpurun_out/pf
# ---- 4. the exact strip scheme's table at the bench's own 512 spp, with the MEASURED exchange (profiles/r06_halo_host_cost.txt)
for mesh in icosphere clustered; do MIRRES_MESH=$mesh timeout -k 5 900 python3 scripts/dev_strip_table.py 512 1 default 2,4,8 3 2>&1 | grep -v amdgpu.ids > $O/${R}_strip_table_512_$mesh.txt; done
cut -c1-700 $O/${R}_bench_default.json; echo; head -8 $O/${R}_kernel_stats_icosphere.csv | cut -c1-150
