# round 4 soak of the persistent base pack (through gpurun): the fuzz fits against the oracle with the pack serving every
# batch (CHB_TILE_SKIP=0) and in the default configuration, then the GPU suite
O=gpurun_out
CHB_TILE_SKIP=0 timeout -k 10 400 python tools/fuzz_fit.py 150 41 > $O/${1}_fuzz_a.txt 2>&1; echo "fuzz a rc=$?"; tail -2 $O/${1}_fuzz_a.txt
CHB_TILE_SKIP=0 timeout -k 10 300 python tools/fuzz_fit.py 50 42 big > $O/${1}_fuzz_b.txt 2>&1; echo "fuzz big rc=$?"; tail -2 $O/${1}_fuzz_b.txt
CHB_TILE_SKIP=0 timeout -k 10 300 python tools/fuzz_fit.py 50 43 m16 > $O/${1}_fuzz_c.txt 2>&1; echo "fuzz m16 rc=$?"; tail -2 $O/${1}_fuzz_c.txt
timeout -k 10 300 python tools/fuzz_fit.py 100 44 > $O/${1}_fuzz_d.txt 2>&1; echo "fuzz default rc=$?"; tail -2 $O/${1}_fuzz_d.txt
python -m pytest tests -m gpu -x -q --durations=8 > $O/${1}_tests.txt 2>&1; echo tests rc=$?; tail -3 $O/${1}_tests.txt
