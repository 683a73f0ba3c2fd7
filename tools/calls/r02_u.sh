#!/bin/bash
# round 2, GPU call U: HEAD sanity after the short-K/V attention kernel: smoke(), the golden engine tests, a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/r02_u_smoke.log
cat $O/r02_u_smoke.log
( timeout 400 python -m pytest tests/test_engine_gpu.py -q -x -k "golden or matches_reference or bf16" 2>&1 | tail -4 ) > $O/r02_u_engine_tests.log
tail -2 $O/r02_u_engine_tests.log
( timeout 300 python bench.py --no-cpu-baseline --no-extra ) > $O/r02_u_bench_short.json 2> $O/r02_u_bench_short.err
cut -c1-330 $O/r02_u_bench_short.json
