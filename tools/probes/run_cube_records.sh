set -x
python -m pytest tests/test_gpu_parity.py tests/test_hex.py tests/test_gpu_general_numbering.py -q -x -m gpu -p no:cacheprovider -k "cluster or cube or hex or narrow" 2>&1 | tail -5
for m in 0 1; do
MPCX_CUBE_CLUSTER_RECORDS=$m python bench.py --steps 20 --warmup 3 --no-sub-records --no-shuffled-record > gpurun_out/cube_rec_$m.json 2>gpurun_out/cube_rec_$m.err; tail -2 gpurun_out/cube_rec_$m.err
MPCX_CUBE_CLUSTER_RECORDS=$m python bench.py --cell hex --steps 20 --warmup 3 --no-sub-records --no-shuffled-record > gpurun_out/hex_rec_$m.json 2>gpurun_out/hex_rec_$m.err; tail -2 gpurun_out/hex_rec_$m.err
done
python tools/show_bench.py gpurun_out/cube_rec_0.json gpurun_out/cube_rec_1.json gpurun_out/hex_rec_0.json gpurun_out/hex_rec_1.json 2>&1 | tail -40
