B="python bench.py --steps 10 --warmup 3 --no-extra --no-e2e --cpu-sample 0"
for i in 1 2 3; do
  CHBIN_LIB=$PWD/ch-bin_amd/libchbin_hip_prev.so $B > gpurun_out/r3_59_prev_$i.json 2>gpurun_out/r3_59.err; echo prev; python tools/bench_line.py gpurun_out/r3_59_prev_$i.json
  $B > gpurun_out/r3_59_new_$i.json 2>gpurun_out/r3_59.err; echo new; python tools/bench_line.py gpurun_out/r3_59_new_$i.json
done
