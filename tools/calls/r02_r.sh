#!/bin/bash
# round 2, GPU call R: the whole GPU suite, smoke(), default bench on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -150 ) > $O/r02_r_gpu_tests.log
tail -4 $O/r02_r_gpu_tests.log
( timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $O/r02_r_smoke.log
cat $O/r02_r_smoke.log
( time timeout 900 python bench.py ) > $O/r02_r_bench_default.json 2> $O/r02_r_bench_default.err
tail -3 $O/r02_r_bench_default.err
cut -c1-500 $O/r02_r_bench_default.json
