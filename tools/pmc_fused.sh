#!/bin/bash
# SQ counter passes of gmm_fused_kernel (gmm-train bench), with and without ablations
export PMC_FILTER=gmm_fused
for a in ${ABLS:-0 15}; do
  export AMX_FUSED_ABL=$a
  bash tools/pmc_run.sh fused_a${a}_1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -- python $GRAFT_REPO_ROOT/bench.py --workload gmm-train --no-cpu-baseline --steps 2 --warmup 1
  bash tools/pmc_run.sh fused_a${a}_2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -- python $GRAFT_REPO_ROOT/bench.py --workload gmm-train --no-cpu-baseline --steps 2 --warmup 1
  bash tools/pmc_run.sh fused_a${a}_3 GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT -- python $GRAFT_REPO_ROOT/bench.py --workload gmm-train --no-cpu-baseline --steps 2 --warmup 1
done
cd $GRAFT_REPO_ROOT; for f in gpurun_out/pmc_fused_a*.txt; do echo "== $f"; cat $f; done
