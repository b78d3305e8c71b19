#!/bin/bash
# build time of the private hierarchies, both meshes; kernel trace of the build alone
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/build_time_sah.txt; : > $out
for mesh in icosphere clustered; do for pt in 0 1 2; do echo "build time $mesh private_tree $pt: $(MIRRES_MESH=$mesh MIRRES_PRIVATE_TREE=$pt python3 scripts/dev_build_time.py 2>&1 | tail -2 | tr "\n" " ")" >> $out; done; done
cat $out
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/kb; MIRRES_MESH=clustered MIRRES_PRIVATE_TREE=2 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kb -o k -- python3 scripts/dev_build_time.py > gpurun_out/kb.log 2>&1
find gpurun_out/kb -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv
rows=list(csv.DictReader(open('{}')))
for r in rows[:24]: print('%-50s calls %5s avg %8.1f us total %8.2f ms' % (r['Name'].replace('void mr::','').replace('mr::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
" | tee gpurun_out/r04/build_kernels_clustered.txt
rm -rf gpurun_out/kb
