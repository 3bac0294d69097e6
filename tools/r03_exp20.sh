# 128 x 128 tile at 4 waves per SIMD (128 registers, small spills) vs 3: microbenchmark in one box
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
P=$R/depth-fusion-in-transformer-based-video-object-detection_amd
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_occ3.txt 2>&1
cp $P/dfx/libdfx.so /tmp/libdfx_full.so
sed 's/BM == 128 \&\& BN == 128 ? 3/BM == 128 \&\& BN == 128 ? 4/' $P/csrc/gemm_f32.hip > /tmp/gemm_occ4.hip
(cd $P/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -c /tmp/gemm_occ4.hip -o /tmp/gemm_occ4.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/dfx/libdfx.so /tmp/gemm_occ4.o $(ls *.o | grep -v gemm_f32.o))
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_occ4.txt 2>&1
cp /tmp/libdfx_full.so $P/dfx/libdfx.so
paste -d'|' <(grep -E "M=|Ci=" $O/gemm_occ3.txt | cut -c1-40,70-125) <(grep -E "M=|Ci=" $O/gemm_occ4.txt | cut -c70-125)
