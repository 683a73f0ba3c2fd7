#!/bin/bash
# round 2, GPU call H: the other configurations through bench.py, for the record
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python bench.py --workload control --images-per-gpu 4 --steps 10 --warmup 2 --no-cpu-baseline --no-extra ) > $O/r02_h_bench_control_4images.json 2> $O/r02_h_err1.log
cut -c1-330 $O/r02_h_bench_control_4images.json
( timeout 600 python bench.py --workload lora_ip --lora-mode fused --steps 20 --warmup 3 --no-cpu-baseline --no-extra ) > $O/r02_h_bench_lora_ip_fused.json 2> $O/r02_h_err2.log
cut -c1-330 $O/r02_h_bench_lora_ip_fused.json
( timeout 600 python bench.py --workload lora_ip --images-per-gpu 4 --steps 10 --warmup 2 --no-cpu-baseline --no-extra ) > $O/r02_h_bench_lora_ip_4images.json 2> $O/r02_h_err3.log
cut -c1-330 $O/r02_h_bench_lora_ip_4images.json
