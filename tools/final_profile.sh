set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_r06d.log 2>&1
grep -m1 '^{"metric"' gpurun_out/bench_r06d.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','train_ms_per_step','train_variant','train_fp32class_ms_per_step','cfg4_bf16_forward_ms')})); print(json.dumps(d['roofline'])[:600])"
bash tools/profile_bench.sh r06_d all > gpurun_out/prof_r06c.log 2>&1
tail -5 gpurun_out/prof_r06c.log
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for p in fp16 bf16x3; do
  mkdir -p $R/gpurun_out/prof_train_$p
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$p -o train -- python $R/tools/profile_train.py $p 10 > $R/gpurun_out/prof_train_$p/run.log 2>&1
  python $R/tools/small_kernels.py $R/gpurun_out/prof_train_$p 12 90 > $R/gpurun_out/r06_d_train_kernels_$p.txt
  rm -f $R/gpurun_out/prof_train_$p/*kernel_trace.csv
done
head -12 $R/gpurun_out/r06_d_train_kernels_fp16.txt
cd $R
timeout 600 python tools/soak_trainer.py fp16 150 2>&1 | tail -3
timeout 600 python tools/soak_trainer.py bf16x3 60 2>&1 | tail -2
