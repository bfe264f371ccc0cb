cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
O=gpurun_out/r03e
cat > /tmp/dbg.py <<'PY'
import numpy as np, torch, sys, os
sys.path.insert(0, '.')
from oracle import farnn_oracle as fo
from re2nn_seq_amd import _lib, synth
dev = torch.device('cuda', 0)
rng = np.random.RandomState(3)
V, S, C, B, L = 950, 71, 128, 256, 64
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
h = _lib.create_onehot_ifst(T, W, O, h0, hT, device=0)
xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(lengths).to(dev)
ref = fo.onehot_ifst_scores(T, W, O, h0, hT, x, lengths)
for rep in range(3):
    tags = torch.full((B, L), -7, dtype=torch.int32, device=dev)
    scores = torch.full((B, L, C), -7.0, dtype=torch.float32, device=dev)
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_FULL, tags.data_ptr(), None, scores.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    sc = scores.cpu().numpy()
    bad = np.argwhere(np.abs(sc - ref).max(axis=2) > 0)
    print(os.environ.get('TAGX'), 'rep', rep, 'bad positions', len(bad))
    for b, i in bad[:6]:
        cols = np.nonzero(sc[b, i] != ref[b, i])[0]
        print('  seq', b, 'len', lengths[b], 'pos', i, 'cols', cols[:8], 'got', sc[b, i, cols[:8]], 'want', ref[b, i, cols[:8]])
PY
TAGX=default timeout 60 python /tmp/dbg.py 2>&1 | grep -v amdgpu.ids
TAGX=nofuse FARNN_NOFUSE=1 timeout 60 python /tmp/dbg.py 2>&1 | grep -v amdgpu.ids | head -3
TAGX=spin0 FARNN_FUSE_SPIN=0 timeout 60 python /tmp/dbg.py 2>&1 | grep -v amdgpu.ids | head -12
