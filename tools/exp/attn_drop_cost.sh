set -e
for m in split bf16; do for p in 0.1 0.0; do echo "== mode $m p $p"; ATTN_MODE=$m ATTN_P=$p python tools/exp/attn_probe.py; done; done
