"""Wire format of proto::HybridGridTSDF checked against the protobuf runtime:
the message type is built from the field list of mapping/proto/3d/hybrid_grid_tsdf.proto:19-31."""
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu


def hybrid_grid_tsdf_message():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "hybrid_grid_tsdf_test.proto"
    fd.package = "cartographer.mapping.proto"
    fd.syntax = "proto3"
    m = fd.message_type.add()
    m.name = "HybridGridTSDF"
    T = descriptor_pb2.FieldDescriptorProto
    for name, num, typ, rep in (("resolution", 1, T.TYPE_FLOAT, False), ("x_indices", 3, T.TYPE_SINT32, True),
                                ("y_indices", 4, T.TYPE_SINT32, True), ("z_indices", 5, T.TYPE_SINT32, True),
                                ("values_tsd", 6, T.TYPE_INT32, True), ("values_weight", 7, T.TYPE_INT32, True),
                                ("relative_truncation_distance", 8, T.TYPE_FLOAT, False),
                                ("max_weight", 9, T.TYPE_FLOAT, False)):
        f = m.field.add()
        f.name, f.number, f.type = name, num, typ
        f.label = T.LABEL_REPEATED if rep else T.LABEL_OPTIONAL
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName("cartographer.mapping.proto.HybridGridTSDF")
    try:
        return message_factory.GetMessageClass(desc)
    except AttributeError:
        return message_factory.MessageFactory(pool).GetPrototype(desc)


def test_to_proto_parses_and_round_trips(po, hg, ctx):
    Msg = hybrid_grid_tsdf_message()
    g = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 14)
    og = po.Grid(0.1)
    for k in range(2):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, 16, 300, stream=k))
        hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), g)
        og.insert(pose[:3], loc)
    data = g.ToProto()
    msg = Msg()
    msg.ParseFromString(data)
    ijk, t, w = og.export()                      # reference iteration order, raw codes incl. marker
    assert msg.resolution == np.float32(0.1)
    assert np.array_equal(np.array(msg.x_indices), ijk[:, 0])
    assert np.array_equal(np.array(msg.y_indices), ijk[:, 1])
    assert np.array_equal(np.array(msg.z_indices), ijk[:, 2])
    assert np.array_equal(np.array(msg.values_tsd), t.astype(np.int64))
    assert np.array_equal(np.array(msg.values_weight), w.astype(np.int64))
    assert msg.relative_truncation_distance == np.float32(0.25)   # sic: ToProto stores getMaxTSD()
    assert msg.max_weight == 1000.0
    assert msg.SerializeToString() == data                         # byte-identical to the runtime
    # proto constructor: same cells and codes back (decode + encode with the loaded converter)
    g2 = hg.HybridGridTSDF.FromProto(ctx, data, max_blocks=1 << 14)
    a, b = g.export(), g2.export()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # the loaded grid decodes with ITS converter: the proto constructor passes field 8 (getMaxTSD() of
    # the writer, 0.025 m here) on as the relative truncation distance (hybrid_grid_tsdf.h:69-83)
    f = np.float32
    writer_max_tsd = f(2.5) * f(0.1)
    assert g2.max_tsd == f(writer_max_tsd * f(0.1)) and g2.min_tsd == -g2.max_tsd
    assert g2.max_weight == f(1000.0)
    cells = a[0][:50]
    tsd = g2.GetTSD(cells)
    assert tsd.dtype == np.float32 and np.all(np.abs(tsd) <= g2.max_tsd + 1e-7)
    assert np.all(g2.GetWeight(cells) > 0) and g2.IsKnown(cells).all()
    # and bytes written by the protobuf runtime load as well
    g3 = hg.HybridGridTSDF.FromProto(ctx, msg.SerializeToString(), max_blocks=1 << 14)
    assert all(np.array_equal(x, y) for x, y in zip(a, g3.export()))
    empty = hg.HybridGridTSDF(ctx, 0.2, max_blocks=64)
    m2 = Msg()
    m2.ParseFromString(empty.ToProto())
    assert len(m2.x_indices) == 0 and m2.resolution == np.float32(0.2)
