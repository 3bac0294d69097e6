# what the GEMM epilogue costs: the microbenchmark with the shipped library and with a build whose tiles end after the K loop
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_epi_full.txt 2>&1
P=$R/depth-fusion-in-transformer-based-video-object-detection_amd
cp $P/dfx/libdfx.so /tmp/libdfx_full.so
(cd $P/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DDFX_GEMM_ABLATE_EPILOGUE -c gemm_f32.hip -o /tmp/gemm_abl.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/dfx/libdfx.so /tmp/gemm_abl.o $(ls *.o | grep -v gemm_f32.o))
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_epi_ablated.txt 2>&1
cp /tmp/libdfx_full.so $P/dfx/libdfx.so
paste -d'|' <(cut -c1-46,95-140 $O/gemm_epi_full.txt) <(cut -c95-140 $O/gemm_epi_ablated.txt)
