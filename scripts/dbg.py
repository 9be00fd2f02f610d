import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
from hectorgrapher_amd import api, synth
import pyoracle as po
ctx=api.Context(0)
g=api.HybridGridTSDF(ctx,0.1,max_blocks=1<<14); og=po.Grid(0.1)
pose=synth.pose_k(0); pts=synth.generate_scan(pose,8,64)
api.TSDFRangeDataInserter3D().Insert(api.RangeData([0,0,0],pts),g); og.insert([0,0,0],pts)
p=api.Problem(ctx); p.add_pose(synth.perturbation()); p.add_block(pts,[g],0.1,0)
c,r,gr,H=p.evaluate()
o=po.Problem(); o.add_pose(synth.perturbation()); o.add_block(pts,[og],0.1,0)
c0,r0,J0,g0=o.evaluate()
print('cost',c,c0); print('r',np.abs(r-r0).max()); print('g',gr,g0); print('H',H[0],(J0.T@J0)[0])
