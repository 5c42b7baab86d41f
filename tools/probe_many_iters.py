"""GPU probe: replay iterations of the shared chain (rsb_submit_many / rsb_collect_many) per channel, as bench.py drives it."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readsb_amd
from bench import make_segment

seg = make_segment(1 << 23, 1)
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iq = np.concatenate([seg] * (per * nch))
nc = per << 23
dev = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
st = torch.cuda.Stream()
torch.cuda.synchronize()
chans = [readsb_amd.Channel(fmt="UC8", fix_bits=1, max_batch_samples=nc) for _ in range(nch)]
outs = [torch.empty((1 << 18) * per * 56, dtype=torch.uint8, pin_memory=True).numpy().view(readsb_amd.MSG_DTYPE) for _ in range(nch)]
for step in range(6):
    for c in chans:
        c.reset()
    readsb_amd.submit_many(chans, [dev.data_ptr() + 2 * nc * i for i in range(nch)], [nc] * nch, st.cuda_stream)
    got = readsb_amd.collect_many(chans, outs)
    print("step", step, "msgs", sorted(set(len(g) for g in got)), "iters", [c.timing()["replay_iterations"] for c in chans], flush=True)
for c in chans:
    c.close()
