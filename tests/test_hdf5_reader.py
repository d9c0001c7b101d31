"""features.hdf without h5py: autolabel_amd/utils/hdf5.py against files written by real h5py / libhdf5
(tests/golden/make_hdf5_fixtures.py, the reference's own writer calls: scripts/compute_feature_maps.py:82-85,160-163)."""
import glob
import json
import os
import pickle
import shutil

import numpy as np
import pytest

from autolabel_amd.utils import hdf5

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'hdf5')
FILES = sorted(glob.glob(os.path.join(HERE, '*.hdf')))


def test_fixtures_present():
    assert len(FILES) == 11


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(p) for p in FILES])
def test_every_dataset_decodes_bit_exact(path):
    want = np.load(path + '.expected.npz')
    meta = json.load(open(path + '.json'))
    with hdf5.File(path, 'r') as f:
        assert f.superblock_version == (3 if meta['libver'] == 'latest' else 0)
        group = f['features']
        assert sorted(group.keys()) == sorted(meta['datasets'])
        for key, m in meta['datasets'].items():
            d = f[f'features/{key}']
            assert list(d.shape) == m['shape'] and d.dtype == np.dtype(m['dtype'])
            assert (list(d.chunks) if d.chunks else None) == m['chunks'] and d.compression == m['compression']
            got = d[:]
            assert got.dtype == want[key].dtype.newbyteorder('=') and got.shape == want[key].shape
            assert np.array_equal(got.view(np.uint8), want[key].astype(got.dtype).view(np.uint8))     # bit for bit
            n = d.shape[0]
            assert np.array_equal(d[1:n - 1], want[key][1:n - 1])
            assert np.array_equal(d[n - 1], want[key][n - 1]) and np.array_equal(d[-1], want[key][-1])
            assert np.array_equal(np.asarray(d), want[key])
            with pytest.raises(IndexError):
                d[n]


def test_reference_attributes():
    """dataset.attrs['pca'|'min'|'range'] (compute_feature_maps.py:116-118; read at autolabel/backend.py:85)."""
    for name in ('ref_latest.hdf', 'ref_earliest.hdf'):
        with hdf5.File(os.path.join(HERE, name)) as f:
            a = f['features/dino'].attrs
            assert pickle.loads(a['pca'].tobytes()) == {'components': 3}
            assert np.array_equal(a['min'], [-1.5, -2.0, -3.25]) and np.array_equal(a['range'], [2.0, 4.0, 6.5])
            assert a['min'].dtype == np.float64


def test_index_kinds_are_all_exercised():
    kinds = {}
    for path in FILES:
        with hdf5.File(path) as f:
            for key in f['features'].keys():
                d = f['features'][key]
                kinds.setdefault(d._index[0] if d._index else d._layout[0], []).append(os.path.basename(path))
    assert {'farray', 'btree1', 'single', 'contiguous'} <= set(kinds)
    with hdf5.File(os.path.join(HERE, 'paged_latest.hdf')) as f:      # > 1024 entries: paged data block
        d = f['features/dino']
        assert int(np.prod(d._chunk_grid())) > 1024 and len(d._chunk_table()) == int(np.prod(d._chunk_grid()))
    with hdf5.File(os.path.join(HERE, 'ref_latest.hdf')) as f:        # incompressible chunks carry the skip-filter mask
        masks = {m for _, _, m in f['features/lseg']._chunk_table().values()}
        assert masks == {1}
        masks = {m for _, _, m in f['features/dino']._chunk_table().values()}
        assert masks == {0}


def test_lzf_c_helper_matches_python_decoder():
    """Every LZF chunk of the fixtures through both decoders (the C one is `aln_lzf_decompress` of the C-ABI library)."""
    from autolabel_amd import hip
    hip.lib()
    n = 0
    for path in FILES:
        with hdf5.File(path) as f:
            for key in f['features'].keys():
                d = f['features'][key]
                if d.compression != 'lzf':
                    continue
                for addr, size, mask in list(d._chunk_table().values())[:40]:
                    if mask & 1:
                        continue
                    raw = bytes(f._r.buf[addr:addr + size])
                    a = hdf5.lzf_decompress(raw, d._chunk_bytes())
                    b = hdf5.lzf_decompress_py(raw, d._chunk_bytes())
                    assert a == b
                    n += 1
                    with pytest.raises(hdf5.Hdf5FormatError):
                        hdf5.lzf_decompress(raw[:-1], d._chunk_bytes())
                    with pytest.raises(hdf5.Hdf5FormatError):
                        hdf5.lzf_decompress(raw, d._chunk_bytes() - 1)
    assert n > 50


def test_lzf_overlapping_reference_known_answer():
    # literal 'ab', then a back reference of length 9 at distance 2 -> 'ab' * 5 + 'a'  (ctrl = 7<<5 | 0, extra len 0, dist-1 = 1)
    stream = bytes([1, ord('a'), ord('b'), (7 << 5), 0, 1])
    assert hdf5.lzf_decompress_py(stream, 11) == b'ababababab' + b'a'
    assert hdf5.lzf_decompress(stream, 11) == b'ababababab' + b'a'


def test_errors():
    with pytest.raises(NotImplementedError):
        hdf5.File(FILES[0], 'w')
    with hdf5.File(os.path.join(HERE, 'ref_latest.hdf')) as f:
        with pytest.raises(KeyError):
            f['features/missing']
        assert 'features/dino' in f and 'features/nope' not in f


def test_not_hdf5(tmp_path):
    p = tmp_path / 'x.hdf'
    p.write_bytes(b'not an hdf5 file at all' * 100)
    with pytest.raises(hdf5.Hdf5FormatError):
        hdf5.File(str(p))


def test_load_features_from_hdf(tmp_path):
    """SceneDataset._load_features (autolabel/dataset.py:438-449) on a scene directory holding only features.hdf."""
    from autolabel_amd import dataset as D
    shutil.copy(os.path.join(HERE, 'ref_latest.hdf'), tmp_path / 'features.hdf')
    want = np.load(os.path.join(HERE, 'ref_latest.hdf.expected.npz'))['dino']

    class Holder(D.SceneDataset):
        def __init__(self, path):                      # only what _load_features touches
            self.scene = type('S', (), {'path': str(path)})()
            self.camera = type('C', (), {'size': (32, 24)})()

    h = Holder(tmp_path)
    h._load_features('dino')
    N, H, W, C = want.shape
    assert h.features.shape == (N, H * W, C) and h.features.dtype == np.float16
    assert np.array_equal(h.features, want.reshape(N, H * W, C))
    assert (h.feature_width, h.feature_height, h.feature_dim) == (W, H, C)
    assert np.array_equal(h._scale_to_feature_xy(np.array([[31.0, 23.0]])), [[W * 31 // 32, H * 23 // 24]])


def test_corrupt_files_raise_cleanly(tmp_path):
    """Random byte damage in the metadata region: the reader returns the data, or raises an ordinary exception -- it must not hang,
    loop, or allocate without bound (cycles of continuation blocks / B-tree nodes, absurd shapes)."""
    import time
    ok = (hdf5.Hdf5FormatError, KeyError, NotImplementedError, IndexError, ValueError, OverflowError, zlib_error(), TypeError,
          AssertionError, MemoryError, UnicodeDecodeError)
    rng = np.random.default_rng(7)
    n_raised = n_read = 0
    for name in ('ref_latest.hdf', 'ref_earliest.hdf', 'gzip_latest.hdf', 'paged_latest.hdf'):
        data = bytearray(open(os.path.join(HERE, name), 'rb').read())
        for trial in range(60):
            bad = bytearray(data)
            for _ in range(int(rng.integers(1, 6))):
                pos = int(rng.integers(8, min(len(bad), 6000)))         # superblock tail, object headers, B-tree / index nodes
                bad[pos] = int(rng.integers(0, 256))
            p = tmp_path / f'{name}.{trial}'
            p.write_bytes(bytes(bad))
            t = time.time()
            try:
                with hdf5.File(str(p)) as f:
                    for key in f['features'].keys():
                        f['features'][key][:]
                n_read += 1
            except ok:
                n_raised += 1
            assert time.time() - t < 20.0
    assert n_raised > 10 and n_read > 10


def zlib_error():
    import zlib
    return zlib.error
