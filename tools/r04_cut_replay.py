import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audiality2_amd
from audiality2_amd.replay import Trace, Backend, replay, read_pcm
tr = Trace(os.path.join(ROOT, "tools/_l5/cut.trace"))
cfg = tr.config
want = read_pcm(os.path.join(ROOT, "tools/_l5/cut.pcm"), 2, 64)
which = sys.argv[1] if len(sys.argv) > 1 else "gpu"
for batch in (1, 64):
    if which == "gpu":
        be = Backend(audiality2_amd.load_library(), "a2amd_", cfg["samplerate"], cfg["basepitch"], cfg["channels"], 0, batch)
    else:
        be = Backend(ctypes.CDLL(os.path.join(ROOT, "oracle", "liba2oracle.so")), "a2o_", cfg["samplerate"], cfg["basepitch"], cfg["channels"])
    out = replay(tr, be, batch=batch)
    be.close()
    n = min(out.shape[1], want.shape[1])
    bad = np.argwhere(out[:, :n] != want[:, :n])
    print(which, "batch", batch, "frames", n, "differ", len(bad), "first", bad[:1], flush=True)
