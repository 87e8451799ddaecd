"""On the GPU box: the JTA train step in both precisions (bench.jta_leg) -> samples/s, ms per step."""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
import bench
d = bench.jta_leg(torch.device("cuda", 0))
print(json.dumps({"fp32": {k: d[k] for k in ("value", "ms_per_step")}, "fp32_gemm": d["roofline"].get("gemm_ms_per_step"),
                  "bf16": d.get("bf16") or d.get("bf16_operands")}))
