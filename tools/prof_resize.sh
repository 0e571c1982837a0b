cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/rz
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/rz/pmc_SQ -o run -- python3 $R/tools/run_kernels.py 3 resize > $R/gpurun_out/rz/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/rz/pmc_LDS -o run -- python3 $R/tools/run_kernels.py 3 resize > $R/gpurun_out/rz/lds.log 2>&1
ls $R/gpurun_out/rz/*
