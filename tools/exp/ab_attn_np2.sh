# A/B on one box: fp32-class JTA train step with the split-mode attention backward on three pieces per operand (round 5) / two (round 6 default)
for rep in 1 2; do for n in 3 2; do EMLOCO_ATTN_BWD_PIECES=$n python tools/exp/jta_step.py 8 2>/dev/null | tail -1 | sed "s/^/attention backward pieces=$n: /"; done; done
for n in 3 2; do echo "== kernels alone (tools/exp/attn_probe.py, split mode), backward pieces $n"; EMLOCO_ATTN_BWD_PIECES=$n ATTN_MODE=split python tools/exp/attn_probe.py 2>&1 | grep bwd; done
for n in 3 2; do echo "== gradients of the shipped-depth model against the reference's (tools/exp/graderr.py), backward pieces $n"; EMLOCO_ATTN_BWD_PIECES=$n python tools/exp/graderr.py 2>&1 | grep -v "amdgpu.ids\|best / second"; done
