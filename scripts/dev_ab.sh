# A/B of library variants under ab/ on the shadow-ray microbenchmark: scripts/dev_ab.sh [mode] variant...
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mode=${1:-0}; shift
for i in 1 2; do
  python3 scripts/dev_any_pmc.py 1600 7 10 $mode 2>&1 | tail -1 | sed 's/^/base  /'
  for v in "$@"; do MIRRES_LIB=$PWD/ab/libmirres_$v.so python3 scripts/dev_any_pmc.py 1600 7 10 $mode 2>&1 | tail -1 | sed "s/^/$v  /"; done
done
