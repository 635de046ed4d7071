cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 1 2 3; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_pw_$m -- $R/tools/pwbench/pwbench 5 0 $m > $R/gpurun_out/pmc_pw_$m.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_pw_b1 -- $R/tools/pwbench/pwbench 5 0 1 > $R/gpurun_out/pmc_pw_b1.log 2>&1
ls -R $R/gpurun_out/pmc_pw_1 | head
