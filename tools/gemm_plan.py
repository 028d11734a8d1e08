"""What every grouped GEMM call of one training step turns into (which problems share the 64 x 64 launch, which the
32 x 32 one, which are launched alone), and every single-problem call: python tools/gemm_plan.py [workload] 2> plan.txt
(the library prints one `[gemm plan]` line per mesm_gemm_group call when MESM_GEMM_PLAN_LOG is set)"""
import os, sys
os.environ.setdefault("MESM_AUTOGRAPH", "0")  # this tool looks at the EAGER step (autograph.py would replay graphs behind these calls)
os.environ.setdefault("MESM_GEMM_PLAN_LOG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.graphed import GraphedStep
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(1234)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
g = GraphedStep(model, crit, batch, args.dataset_name, warmup=0)  # capture = one pass through every launch site
torch.cuda.synchronize()
