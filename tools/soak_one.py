import sys, os
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
from tests import soak_fuzz as S
for seed in [int(x) for x in sys.argv[1:]]:
    try:
        tag, w = S.one(seed)
        print("ok", w, tag, flush=True)
    except Exception as e:
        print("EXC", seed, repr(e)[:300], flush=True)
