import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from hectorgrapher_amd import api, synth
from hectorgrapher_amd import distributed as hgd
dev = torch.device("cuda", 0)
ctx = api.Context(0)
for res in (0.05, 0.1, 0.2):
    g = api.HybridGridTSDF(ctx, res, max_blocks=1 << 14)
    ins = [api.TSDFRangeDataInserter3D()]
    for k in range(3):
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, 16, 400, stream=k)
        api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), [g], pose_tq=pose.astype(np.float32))
    ctx.synchronize()
    nb = g.num_blocks()
    keys, vox = hgd.grid_block_tensors(g, dev)
    kc, vc = keys.cpu(), vox.cpu()
    print(res, "nb", nb, keys.shape, vox.shape, "distinct keys", len(set(kc.tolist())))
    a = g.export()
    for mb in (max(64, nb), 1 << 14):
        fresh = api.HybridGridTSDF(ctx, res, max_blocks=mb)
        fresh.import_blocks(kc.numpy().view(np.uint64), vc.numpy().view(np.uint32).reshape(-1))
        b = fresh.export()
        print("  import max_blocks", mb, "nb", fresh.num_blocks(), "voxels", len(a[1]), len(b[1]),
              "equal", all(np.array_equal(x, y) for x, y in zip(a, b)))
        if len(a[1]) == len(b[1]) and not np.array_equal(a[0], b[0]):
            d = np.nonzero((a[0] != b[0]).any(1))[0]
            print("   first ijk diff", d[:5], a[0][d[:3]], b[0][d[:3]])
        fresh.close()
    # voxel sums
    print("  nonzero voxels in packed", int((vc != 0).sum()), "export", len(a[1]))
