"""Dictionary decode on the GPU (C ABI zsmi_decompress_usingDict / zsmi_decompressBatchHost_usingDict): frames
that upstream libzstd compressed with a raw-content and with a trained dictionary (tests/golden/gen_fixtures_dict.py) must come
back byte for byte, and the failures must be the reference's (ZStdDecompress.cs:632-634, :2378-2450), as under oracle D."""
import ctypes, os
import numpy as np
import pytest
import _oracle as O
import _data as D

pytestmark = pytest.mark.gpu
FIX = np.load(os.path.join(D.GOLDEN, "libzstd_fixtures_dict.npz"))
NAMES = sorted(k[:-6] for k in FIX.files if k.endswith("_frame"))
ERR = 0xFFFFFF88


@pytest.fixture(scope="module")
def codec():
    from zstandard_amd import BatchCodec
    return BatchCodec(0)


def _decode(codec, frames, caps, dictionary):
    sizes = np.array([len(f) for f in frames], dtype=np.uint32)
    offs = np.zeros(len(frames), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    out, oo, osz = codec.decompress_host(blob, offs, sizes, np.array(caps, dtype=np.uint32), dictionary)
    return [(int(osz[i]), out[int(oo[i]):int(oo[i]) + (int(osz[i]) if osz[i] <= ERR else 0)].tobytes()) for i in range(len(frames))]


@pytest.mark.parametrize("kind", ["raw", "trained"])
def test_dictionary_frames_decode_in_one_batch(codec, kind):
    names = [n for n in NAMES if n.startswith(kind)]
    dic = FIX[names[0] + "_dict"].tobytes()
    frames = [FIX[n + "_frame"].tobytes() for n in names]; wants = [FIX[n + "_want"].tobytes() for n in names]
    for (sz, got), want in zip(_decode(codec, frames, [len(w) for w in wants], dic), wants):
        assert sz == len(want) and got == want


def test_one_shot_call_with_dictionary(codec):
    Z = codec.L
    for name in ("raw_small_l3", "trained_small_l3", "trained_two_blocks_l3"):
        dic, frame, want = FIX[name + "_dict"].tobytes(), FIX[name + "_frame"].tobytes(), FIX[name + "_want"].tobytes()
        out = ctypes.create_string_buffer(len(want))
        r = Z.zsmi_decompress_usingDict(out, len(want), frame, len(frame), dic, len(dic))
        assert r == len(want) and out.raw == want
        assert Z.zsmi_decompress_usingDict(out, len(want), frame, len(frame), None, 0) > ERR      # the same frame without its dictionary


def test_dictionary_errors_are_the_references(codec):
    name = "trained_small_l3"
    dic, frame, want = FIX[name + "_dict"].tobytes(), FIX[name + "_frame"].tobytes(), FIX[name + "_want"].tobytes()
    cases = [(b"", 32), (FIX["raw_small_l3_dict"].tobytes(), 32), (dic[:9], 30), (dic[:40], 30), (dic[:120], 30)]
    other = bytearray(dic); other[4] ^= 1; cases.append((bytes(other), 32))
    for d, code in cases:
        (sz, _), = _decode(codec, [frame], [len(want)], d)
        assert sz > ERR and (0x100000000 - sz) == code
        with pytest.raises(O.OracleError) as e:
            O.decompress_using_dict(frame, len(want), d) if d else O.decompress(frame, len(want))
        assert e.value.code == code
    # raw-content frames without their dictionary: an offset reaches in front of the output (or the bytes differ)
    rname = "raw_text_l19"
    (sz, got), = _decode(codec, [FIX[rname + "_frame"].tobytes()], [len(FIX[rname + "_want"])], b"")
    assert sz > ERR or got != FIX[rname + "_want"].tobytes()


def test_own_frames_still_decode_when_a_dictionary_is_given(codec):
    """a frame that names no dictionary decodes the same with any raw-content dictionary loaded (the general kernel is used)"""
    data = D.zipf_log(1 << 18, seed_lo=3)
    chunks = [data[i * 40000:(i + 1) * 40000].tobytes() for i in range(4)]
    frames = [O.compress(c, 3) for c in chunks]
    for (sz, got), want in zip(_decode(codec, frames, [len(c) for c in chunks], FIX["raw_small_l3_dict"].tobytes()), chunks):
        assert sz == len(want) and got == want
