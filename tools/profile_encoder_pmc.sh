#!/bin/bash
# instruction mix (rocprofv3 PMC) of the encoder kernels on a query batch and a passage batch; run on the GPU box
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/encpmc_q -- python3 $R/tools/profile_query_encoder.py > /dev/null 2>&1
ENC_N=64 ENC_L=300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/encpmc_p -- python3 $R/tools/profile_query_encoder.py > /dev/null 2>&1
ls $R/gpurun_out/encpmc_q/*/ $R/gpurun_out/encpmc_p/*/
