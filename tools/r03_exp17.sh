set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_conv_gpu.py tests/test_models_gpu.py tests/test_detector_golden.py -x -q -m gpu > $O/t_exp17.log 2>&1 || { tail -30 $O/t_exp17.log; exit 1; }
tail -1 $O/t_exp17.log
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_epi_new4.txt 2>&1
DFX_GEMM_OLD_EPILOGUE=1 FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_epi_old4.txt 2>&1
paste -d'|' <(grep -E "M=|Ci=" $O/gemm_epi_old4.txt | cut -c1-40,95-125) <(grep -E "M=|Ci=" $O/gemm_epi_new4.txt | cut -c95-125) | tail -10
grep -E "M=" $O/gemm_epi_old4.txt | cut -c1-40,70-120; grep -E "M=" $O/gemm_epi_new4.txt | cut -c1-40,70-120
python tools/gemm_in_step.py 32 > $O/gemm_in_step_new2.txt 2>&1; head -3 $O/gemm_in_step_new2.txt
python bench.py --no-cpu-baseline > $O/bench4.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in d['roofline_kernels']: print(k['kernel'][:40], k['frac'], k.get('ms_per_step'))"
