"""dataset/dataset_common.py: TFRecord framing + tf.Example wire format + the slim decoders, without TensorFlow.
The wire parser is pinned against the protobuf runtime itself: a tf.Example message type is declared at run time (same field numbers
as tensorflow/core/example/{example,feature}.proto) and ITS serialisation — packed and unpacked repeated fields — must parse to the same
values; the writer of convert_tfrecords.py's schema must in turn parse with the protobuf runtime."""
import io

import numpy as np
import pytest

from dan_amd.dataset import dataset_common as DC


def _example_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="ex_test.proto", package="ext", syntax="proto3")
    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m
    F = descriptor_pb2.FieldDescriptorProto
    b = msg("BytesList"); f = b.field.add(name="value", number=1, type=F.TYPE_BYTES, label=F.LABEL_REPEATED)
    fl = msg("FloatList"); f = fl.field.add(name="value", number=1, type=F.TYPE_FLOAT, label=F.LABEL_REPEATED)
    il = msg("Int64List"); f = il.field.add(name="value", number=1, type=F.TYPE_INT64, label=F.LABEL_REPEATED)
    ft = msg("Feature")
    ft.oneof_decl.add(name="kind")
    ft.field.add(name="bytes_list", number=1, type=F.TYPE_MESSAGE, type_name=".ext.BytesList", label=F.LABEL_OPTIONAL, oneof_index=0)
    ft.field.add(name="float_list", number=2, type=F.TYPE_MESSAGE, type_name=".ext.FloatList", label=F.LABEL_OPTIONAL, oneof_index=0)
    ft.field.add(name="int64_list", number=3, type=F.TYPE_MESSAGE, type_name=".ext.Int64List", label=F.LABEL_OPTIONAL, oneof_index=0)
    fs = msg("Features")
    ent = fs.nested_type.add(name="FeatureEntry")
    ent.options.map_entry = True
    ent.field.add(name="key", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    ent.field.add(name="value", number=2, type=F.TYPE_MESSAGE, type_name=".ext.Feature", label=F.LABEL_OPTIONAL)
    fs.field.add(name="feature", number=1, type=F.TYPE_MESSAGE, type_name=".ext.Features.FeatureEntry", label=F.LABEL_REPEATED)
    ex = msg("Example")
    ex.field.add(name="features", number=1, type=F.TYPE_MESSAGE, type_name=".ext.Features", label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("ext.Example"))


def test_wire_parser_against_the_protobuf_runtime():
    Example = _example_classes()
    e = Example()
    e.features.feature["image/filename"].bytes_list.value.append(b"0--Parade/0_Parade_marchingband_1_849.jpg")
    e.features.feature["image/object/bbox/xmin"].float_list.value.extend([0.125, 0.5, 0.33])
    e.features.feature["image/shape"].int64_list.value.extend([768, 1024, 3])
    e.features.feature["image/object/bbox/pose"].int64_list.value.extend([0, 1, -1, 1 << 40])
    e.features.feature["empty"].float_list.SetInParent()
    got = DC.parse_example(e.SerializeToString())
    assert got["image/filename"] == [b"0--Parade/0_Parade_marchingband_1_849.jpg"]
    assert np.array_equal(got["image/object/bbox/xmin"], np.asarray([0.125, 0.5, 0.33], np.float32))
    assert got["image/shape"].tolist() == [768, 1024, 3]
    assert got["image/object/bbox/pose"].tolist() == [0, 1, -1, 1 << 40]
    assert len(got["empty"]) == 0
    # and the other way round: what convert_to_example writes is a valid tf.Example for the protobuf runtime
    ser = DC.convert_to_example("a.jpg", b"\xff\xd8jpegbytes", [[0.1, 0.2, 0.3, 0.4], [0.5, 0.6, 0.7, 0.8]], [0, 2], [1, 0], [0, 0], [0, 1], [2, 0], [1, 1],
                                480, 640)
    m = Example()
    m.ParseFromString(ser)
    f = m.features.feature
    assert list(f["image/shape"].int64_list.value) == [480, 640, 3]
    assert np.allclose(list(f["image/object/bbox/ymin"].float_list.value), [0.1, 0.5]) and np.allclose(list(f["image/object/bbox/xmax"].float_list.value), [0.4, 0.8])
    assert f["image/encoded"].bytes_list.value[0] == b"\xff\xd8jpegbytes" and f["image/format"].bytes_list.value[0] == b"JPEG"
    assert list(f["image/object/bbox/occlusion"].int64_list.value) == [2, 0]


def test_crc32c_known_answers_and_framing(tmp_path):
    assert DC._crc32c(b"123456789") == 0xE3069283                         # the CRC-32C check value (RFC 3720 B.4)
    assert DC._crc32c(b"\x00" * 32) == 0x8A9136AA
    p = str(tmp_path / "a.tfrecord")
    DC.write_tfrecord(p, [b"first", b"", b"x" * 1000])
    assert list(DC.read_tfrecord(p, verify_payload=True)) == [b"first", b"", b"x" * 1000]
    raw = bytearray(open(p, "rb").read())
    raw[14] ^= 1                                                           # flip a payload bit of record 0
    open(p, "wb").write(bytes(raw))
    assert next(DC.read_tfrecord(p)) != b"first"                           # unverified read passes the damage through ...
    with pytest.raises(IOError):
        list(DC.read_tfrecord(p, verify_payload=True))                     # ... the verified one refuses it
    raw[14] ^= 1
    raw[3] ^= 1                                                            # damaged length
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        list(DC.read_tfrecord(p))


def _jpeg(h, w, seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    img = (rng.rand(h // 8 + 1, w // 8 + 1, 3) * 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w]
    b = io.BytesIO()
    Image.fromarray(img).save(b, format="JPEG", quality=95)
    return b.getvalue(), img


def test_slim_get_batch_training_and_evaluation_entries(tmp_path):
    recs = []
    for i in range(10):
        enc, _ = _jpeg(40 + i, 56, i)
        boxes = [] if i == 3 else [[0.1, 0.2, 0.5, 0.6], [0.3, 0.3, 0.9, 0.8]][: 1 + i % 2]
        k = len(boxes)
        recs.append(DC.convert_to_example("img%d.jpg" % i, enc, boxes, [0] * k, [0] * k, [0] * k, [0] * k, [0] * k, [0] * k, 40 + i, 56))
    DC.write_tfrecord(str(tmp_path / "wider_train-00000-of-00001"), recs)
    pattern = str(tmp_path / "wider_{}-*")
    item = DC.decode_record(recs[4])
    assert item["image"].shape == (44, 56, 3) and item["image"].dtype == np.uint8 and item["shape"].tolist() == [44, 56, 3]
    assert item["object/bbox"].shape == (1, 4) and np.allclose(item["object/bbox"][0], [0.1, 0.2, 0.5, 0.6]) and item["filename"] == b"img4.jpg"

    def prep_train(image, bboxes):
        return image[:32, :32].astype(np.float32), bboxes

    def encoder(b):
        return [np.zeros((5, 4), np.float32) + len(b)], [np.ones((5,), np.int64)], [np.zeros((5,), np.float32)], [b]

    seen = []
    for batch in DC.slim_get_batch(2, 3, "train", pattern, 2, 2, prep_train, encoder, num_epochs=1, is_training=True, seed=1):
        assert len(batch) == 3
        for e in batch:
            assert len(e) == 3 + 4 and e[0].shape == (32, 32, 3) and e[3].shape == (5, 4)
            seen.append(e[1])
    assert b"img3.jpg" not in seen and len(seen) == 9 and len(set(seen)) == 9          # the face-less image is skipped, every other one is seen once
    assert seen != sorted(seen)                                                      # shuffled

    def prep_eval(image, bboxes):
        return image.astype(np.float32), np.asarray(image.shape[:2])

    out = list(DC.slim_get_batch(2, 4, "train", pattern, 1, 1, prep_eval, None, num_epochs=1, is_training=False))
    assert [len(b) for b in out] == [4, 4, 1]                                        # file order, smaller final batch
    assert [e[1] for b in out for e in b] == [b"img%d.jpg" % i for i in range(10) if i != 3]
    with pytest.raises(ValueError):
        next(DC.slim_get_batch(2, 4, "test", pattern, 1, 1, prep_eval, None))
