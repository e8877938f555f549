"""Oracle D with a dictionary (ZSTD_decompress_usingDict, ZStdDecompress.cs:2162; LoadEntropy :2378,
RefDictContent :2366, the dictionary-segment match copy :1290-1315), pinned by frames that upstream libzstd compressed with a
raw-content and with a trained dictionary (tests/golden/gen_fixtures_dict.py).  The HIP decoder does not take dictionaries
yet: this is the checker for that row."""
import os
import numpy as np
import pytest
import _oracle as O
import _data as D

FIX = np.load(os.path.join(D.GOLDEN, "libzstd_fixtures_dict.npz"))
NAMES = sorted(k[:-6] for k in FIX.files if k.endswith("_frame"))


@pytest.mark.parametrize("name", NAMES)
def test_dictionary_frames_decode(name):
    dic, frame, want = FIX[name + "_dict"].tobytes(), FIX[name + "_frame"].tobytes(), FIX[name + "_want"].tobytes()
    assert O.decompress_using_dict(frame, len(want), dic) == want


@pytest.mark.parametrize("name", [n for n in NAMES if n.startswith("trained")])
def test_trained_frames_need_their_dictionary(name):
    frame, want = FIX[name + "_frame"].tobytes(), FIX[name + "_want"].tobytes()
    with pytest.raises(O.OracleError) as e:
        O.decompress(frame, len(want))                      # the frame names a dictionary, none is loaded (:632-634)
    assert e.value.code == 32
    with pytest.raises(O.OracleError) as e:
        O.decompress_using_dict(frame, len(want), FIX["raw_small_l3_dict"].tobytes())     # a raw-content dictionary has no ID
    assert e.value.code == 32


def test_raw_content_frames_without_their_dictionary_fail_or_differ():
    for name in [n for n in NAMES if n.startswith("raw")]:
        frame, want = FIX[name + "_frame"].tobytes(), FIX[name + "_want"].tobytes()
        try:
            got = O.decompress(frame, len(want))
        except O.OracleError as e:
            assert e.code == 20                             # an offset reaches in front of the output (:1293)
        else:
            assert got != want


def test_damaged_dictionary_is_reported():
    name = "trained_small_l3"
    dic, frame, want = FIX[name + "_dict"].tobytes(), FIX[name + "_frame"].tobytes(), FIX[name + "_want"].tobytes()
    for cut in (9, 40, 120):                                # inside the entropy tables (LoadEntropy :2378-2450)
        with pytest.raises(O.OracleError) as e:
            O.decompress_using_dict(frame, len(want), dic[:cut])
        assert e.value.code == 30
    bad = bytearray(dic); bad[4] ^= 1                       # another dictID: not the frame's dictionary
    with pytest.raises(O.OracleError) as e:
        O.decompress_using_dict(frame, len(want), bytes(bad))
    assert e.value.code == 32


def test_without_dictionary_the_entry_points_agree():
    frame, want = D.fixtures()["text64k_l3"]
    assert O.decompress_using_dict(frame, len(want), b"") == want == O.decompress(frame, len(want))
