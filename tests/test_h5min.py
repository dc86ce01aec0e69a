"""vamp_amd/h5min.py: the subset of HDF5 behind the reference's file contract
(vpspectrum.py:260-266, 528-538), without h5py.  Pinned by a real h5py-written file of the
reference (tests/golden/simba_H1215.h5, copied by make_golden.py)."""
import os
import struct

import numpy as np
import pytest

from conftest import ROOT, load_golden
from vamp_amd import h5min

REAL = os.path.join(ROOT, "tests", "golden", "simba_H1215.h5")


def test_reads_a_file_written_by_h5py():
    d = h5min.read(REAL)
    assert sorted(d) == ["density_col", "flux", "noise", "tau", "temp", "velocity", "wavelength"]
    g = load_golden("simba_spectra.npz")
    for k in ("wavelength", "flux", "noise"):
        assert d[k].dtype == np.float64 and np.array_equal(d[k], g["H1215_" + k])


def _members(reader):
    _, _, cache, scratch = reader.symbol_entry(reader.root_entry)
    assert cache == 1
    bt, hp = struct.unpack_from("<QQ", scratch)
    return {name: ohdr for name, ohdr, _, _ in reader.group_members(bt, hp)}


def test_writer_reproduces_h5py_structures(tmp_path):
    """The same datasets written by h5min: superblock fields, group B-tree / heap / symbol table
    and every object-header message come out as h5py wrote them, apart from the addresses."""
    real = h5min._Reader(open(REAL, "rb").read())
    data = h5min.read(REAL)
    mine_path = tmp_path / "clone.h5"
    h5min.write(str(mine_path), data, mtime=0x5BACF06D)
    mine = h5min._Reader(open(mine_path, "rb").read())
    assert mine.b[:24] == real.b[:24]                       # signature, versions, sizes, K values, flags
    ma, mb = _members(real), _members(mine)
    assert list(ma) == list(mb)                             # same names in the same (sorted) order
    for name in ma:
        for (ta, fa, da), (tb, fb, db) in zip(real.messages(ma[name]), mine.messages(mb[name])):
            if ta == 0x08:                                  # layout: the data address differs
                da, db = da[:2] + da[10:], db[:2] + db[10:]
            assert (ta, fa, da) == (tb, fb, db), (name, hex(ta))
    back = h5min.read(str(mine_path))
    assert all(np.array_equal(back[k], data[k]) for k in data)


def test_round_trip_of_the_result_file_shapes(tmp_path):
    rng = np.random.default_rng(0)
    data = {"b": rng.normal(size=5), "N": rng.normal(size=(3, 4)), "region_numbers": np.arange(7), "difficult_fit": True,
            "scalar": np.float64(3.5), "f32": rng.normal(size=6).astype(np.float32), "i32": np.arange(4, dtype=np.int32),
            "empty": np.zeros(0), "cube": rng.normal(size=(2, 3, 4)), "region_pixels": np.array([[1, 5], [9, 30]])}
    for i in range(430):                                    # one per region of a long spectrum
        data["region_%d_flux" % i] = rng.normal(size=(1 + i % 3, 9 + i % 50))
    path = tmp_path / "flux_model.h5"
    h5min.write(str(path), data)
    back = h5min.read(str(path))
    assert set(back) == set(data)
    for k, v in data.items():
        v = np.asarray(v)
        assert back[k].shape == v.shape, k
        assert np.array_equal(back[k], v.astype(np.int8) if v.dtype == bool else v), k
    assert back["region_numbers"].dtype == np.int64 and back["f32"].dtype == np.float32 and back["difficult_fit"].dtype == np.int8


def test_rejects_what_it_does_not_understand(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(h5min.H5FormatError):
        h5min.read(str(p))
    buf = bytearray(open(REAL, "rb").read())
    buf[8] = 2                                              # superblock version 2: outside the subset
    p.write_bytes(bytes(buf))
    with pytest.raises(h5min.H5FormatError):
        h5min.read(str(p))
    with pytest.raises(TypeError):
        h5min.write(str(p), {"s": np.array(["text"])})
    with pytest.raises(ValueError):
        h5min.write(str(p), {"a/b": np.zeros(3)})
