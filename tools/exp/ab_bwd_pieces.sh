# A/B on one box: fp32-class JTA train step, pieces per operand of the backward's gradient products (round 6):
#   attention backward (EMLOCO_ATTN_BWD_PIECES), input-gradient GEMMs (EMLOCO_BWD_PIECES_DX), weight-gradient GEMMs (EMLOCO_BWD_PIECES_DW)
run() { EMLOCO_ATTN_BWD_PIECES=$1 EMLOCO_BWD_PIECES_DX=$2 EMLOCO_BWD_PIECES_DW=$3 python tools/exp/jta_step.py ${STEPS:-12} 2>/dev/null | tail -1 | sed "s/^/attention $1  dx $2  dw $3 : /"; }
for rep in 1 2; do run 3 3 3; run 2 3 3; run 2 2 3; run 2 2 2; done
for cfg in "3 3 3" "2 2 3" "2 2 2"; do set -- $cfg; echo "== gradients of the shipped-depth model against the reference's (tools/exp/graderr.py): attention $1, dx $2, dw $3"
  EMLOCO_ATTN_BWD_PIECES=$1 EMLOCO_BWD_PIECES_DX=$2 EMLOCO_BWD_PIECES_DW=$3 python tools/exp/graderr.py 2>&1 | grep -v "amdgpu.ids\|best / second"; done
