# A/B of the software-pipelined split-mode attention backward (EMLOCO_ATTN_PIPE=0: round 5's kernels)
for m in ${MODES:-split}; do for pipe in 0 1 0 1; do echo "== mode $m pipe $pipe"; EMLOCO_ATTN_PIPE=$pipe ATTN_MODE=$m python tools/exp/attn_probe.py 2>&1 | grep -v amdgpu.ids; done; done
