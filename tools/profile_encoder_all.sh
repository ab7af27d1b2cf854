set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/encq -- python3 $R/tools/profile_query_encoder.py > /dev/null 2>&1
ENC_N=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/enc1 -- python3 $R/tools/profile_query_encoder.py > /dev/null 2>&1
ENC_N=64 ENC_L=300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/encp2 -- python3 $R/tools/profile_query_encoder.py > /dev/null 2>&1
cd $R
COLBERT_ENC_GEMM_FORM=1 timeout -k 10 300 python -m pytest tests/test_encoder.py -x -q -m gpu -k "bert_forward or base_shape or f16x3" > gpurun_out/enc_form1.log 2>&1; tail -2 gpurun_out/enc_form1.log
ls gpurun_out/encq/*/ gpurun_out/enc1/*/ gpurun_out/encp2/*/ | head -20
