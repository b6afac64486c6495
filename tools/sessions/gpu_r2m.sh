#!/bin/bash
OUT=gpurun_out/r2m; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc_mfma_panel -o pmc -- python $R/tools/panel_bench.py 12288 16 > $R/$OUT/pmc_mfma_panel.log 2>&1); echo "pmc mfma panel exit $?" | tee -a $OUT/session.log
tail -3 $OUT/pmc_mfma_panel.log | tee -a $OUT/session.log
DBM=$(find $OUT/pmc_mfma_panel -name "*.db" | head -1)
python tools/pmc_dump.py $DBM panel16 | tail -4 | tee -a $OUT/session.log
python tools/mfma_util.py $DBM panel16_mfma_kernel 4831838208 | tee -a $OUT/session.log
rm -rf $OUT/pmc_mfma_panel
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc_mfma_eigh -o pmc -- python $R/tools/eigh_only.py 3072 1 > $R/$OUT/pmc_mfma_eigh.log 2>&1); echo "pmc mfma eigh exit $?" | tee -a $OUT/session.log
DBM=$(find $OUT/pmc_mfma_eigh -name "*.db" | head -1)
python tools/mfma_util.py $DBM wy_apply_mfma_kernel 57982058496 | tee -a $OUT/session.log
python tools/mfma_util.py $DBM gemm128_merge_batched_kernel | tee -a $OUT/session.log
python tools/mfma_util.py $DBM rank2k_stream_kernel | tee -a $OUT/session.log
rm -rf $OUT/pmc_mfma_eigh
