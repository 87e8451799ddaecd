for rep in 1 2; do for v in base p2w4 p2w3; do EMLOCO_LIB=$PWD/variants/full_$v.so python tools/exp/jta_step.py 12 2>/dev/null | tail -1 | sed "s/^/$v: /"; done; done
