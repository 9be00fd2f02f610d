import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (test infrastructure)."""
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hg():
    """The product API; requires the built HIP library and a GPU."""
    from hectorgrapher_amd import api
    return api


@pytest.fixture(scope="session")
def ctx(hg):
    c = hg.Context(0)
    yield c
    c.close()


def build_map(po_or_none, hg_ctx_and_api, resolutions, rings, cols, num_scans, max_blocks=1 << 15,
              opts_kw=None):
    """Inserts `num_scans` synthetic scans at the ground-truth poses into oracle and/or GPU grids."""
    from hectorgrapher_amd import synth
    opts_kw = opts_kw or {}
    ogrids = [po_or_none.Grid(r) for r in resolutions] if po_or_none else None
    ggrids = None
    inserter = None
    if hg_ctx_and_api:
        c, api = hg_ctx_and_api
        per_level = max_blocks if isinstance(max_blocks, (list, tuple)) else [max_blocks] * len(resolutions)
        ggrids = [api.HybridGridTSDF(c, r, max_blocks=m) for r, m in zip(resolutions, per_level)]
        inserter = api.TSDFRangeDataInserter3D(api.InsertOpts(**opts_kw))
    for k in range(num_scans):
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, rings, cols, stream=k)
        loc = synth.transform_points(pose, pts)
        origin = pose[:3].astype(np.float32)
        if ogrids:
            for g in ogrids:
                g.insert(origin, loc, po_or_none.InsertOpts(**opts_kw))
        if ggrids:
            for g in ggrids:
                inserter.Insert(api.RangeData(origin, loc), g)
    return ogrids, ggrids
