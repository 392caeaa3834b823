import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, runpy
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    sys.argv = ["host_turn.py", "300"]
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts/dev/host_turn.py"), run_name="__main__")
