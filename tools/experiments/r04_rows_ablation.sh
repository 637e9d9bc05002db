#!/bin/bash
# where the row pass of the distance transform spends its time: experiment builds of dvo_frames.hip with the packed scan's trips, the d2
# store or the presence-bitmap update compiled out (make EXP=abl_<X> EXPDEFS=-DABL_<X>=1 with temporary #ifdefs at those three places; wrong
# results, timing only) -> profiles/r04_frames/rows_ablation.txt: store 2.6 %, bitmap 9.7 %, the scan + staging the rest (without the
# packed scan every pixel falls to the exact 32-bit finish: 2.6x slower)
for v in "" _abl_NOSCAN _abl_NOSTORE _abl_NOBITMAP; do
  export DVO_LIB_VARIANT=$v
  bash tools/prof_frames.sh > gpurun_out/pf.log 2>&1
  echo "variant '$v':"; grep "edt_rows_pk_levels\|edt_rank_pack_levels_kernel<2048\|edt_columns8_levels" gpurun_out/prof_frames/summary.txt | head -3 | cut -c1-140
done
