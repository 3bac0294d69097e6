set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_conv_gpu.py -x -q -m gpu > $O/t_conv_dpp2.log 2>&1 || { tail -40 $O/t_conv_dpp2.log; exit 1; }
tail -2 $O/t_conv_dpp2.log
for i in 1 2; do FRAMES=32 LIB=0 python tools/bench_conv.py > $O/conv_dpp2_$i.txt 2>&1; grep wino $O/conv_dpp2_$i.txt | sed 's/| dfx\[igemm\].*//'; done
