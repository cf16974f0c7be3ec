import sys, os
sys.argv = [sys.argv[0]] + sys.argv[1:]
import torch
torch.cuda.synchronize()
exec(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/quick_bench.py")).read())
