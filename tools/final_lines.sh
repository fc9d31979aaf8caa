#!/bin/bash
# the un-profiled bench lines of a profiles/ set: default (training step, batch 4), eval forward fp32 / bf16 / fp16
SET=${1:-rXX}
python bench.py > gpurun_out/${SET}_bench_fwdbwd.json.log 2>gpurun_out/${SET}_bench_fwdbwd.err && echo fwdbwd done
python bench.py --mode fwd --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${SET}_bench_fwd_f32.json.log 2>/dev/null && echo fwd done
python bench.py --mode fwd --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${SET}_bench_fwd_bf16.json.log 2>/dev/null && echo bf16 done
python bench.py --mode fwd --dtype fp16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${SET}_bench_fwd_fp16.json.log 2>/dev/null && echo fp16 done
