set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_conv_gpu.py -x -q -m gpu > $O/t_conv_dpp.log 2>&1 || { tail -40 $O/t_conv_dpp.log; exit 1; }
tail -2 $O/t_conv_dpp.log
python tools/wino_stamp.py > $O/wino_stamp2.txt 2>&1; grep -E "conv|iteration|issued|transform" $O/wino_stamp2.txt
FRAMES=32 LIB=0 python tools/bench_conv.py > $O/conv_dpp.txt 2>&1; grep wino $O/conv_dpp.txt | sed 's/| dfx\[igemm\].*//'
python tools/proto_bf16x3.py > $O/proto_bf16x3.txt 2>&1; tail -6 $O/proto_bf16x3.txt
python tools/copy_kernel_origins.py 4 > $O/copy_origins_f4.txt 2>&1; tail -40 $O/copy_origins_f4.txt
