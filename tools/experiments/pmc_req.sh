#!/bin/bash
# L2->fabric read requests per alignment of bench variants: tools/pmc_req.sh <tag> -- one guarded --pmc pass per variant
TAG=$1
for v in "it10:" "it1:--iters 1" "it5:--iters 5" "nofinal:--no-final-outputs" "alias1:--debug-alias 1" "b1024:--batch 1024"; do
  n=${v%%:*}; a=${v#*:}
  tools/pmc.sh ${TAG}_$n "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum" --no-extra-legs $a 2>&1 | grep -v "^rc=" | sed "s/^/$n /"
done
