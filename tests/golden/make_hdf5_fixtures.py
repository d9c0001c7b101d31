"""Writes tests/golden/hdf5/*.hdf with REAL h5py / libhdf5 and the reference's own calls, plus the arrays they must decode to.

Run with an interpreter that has h5py (this container: /opt/conda/bin/python3.9 -> h5py 3.3.0, libhdf5 1.10.6, LZF filter):

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py

`ref_*` files are written exactly as /root/reference/scripts/compute_feature_maps.py does (:160-163 File(..., 'w',
libver='latest') + create_group('features'); :82-85 create_dataset(name, shape, dtype=np.float16, compression='lzf'); :116-118
the 'pca' / 'min' / 'range' attributes); the others vary what a user's h5py could vary (library-version bounds, chunk shapes,
filters, dtype) so that every index / filter branch of autolabel_amd/utils/hdf5.py meets a real file."""
import json
import os
import pickle

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hdf5')
rng = np.random.default_rng(20260210)


def feature_like(shape, dtype=np.float16):
    """Smooth low-rank maps quantised to a few levels: compresses (long LZF back references) like PCA-reduced features."""
    n = int(np.prod(shape[:-1]))
    basis = rng.standard_normal((3, shape[-1]))
    t = np.linspace(0, 6.0, n)[:, None]
    coef = np.concatenate([np.sin(t), np.cos(0.5 * t), np.ones_like(t)], 1)
    a = np.round(coef @ basis * 4) / 4
    a[rng.random(n) < 0.05] = 0.0                       # runs of zeros: overlapping (period-1/2) references
    return a.reshape(shape).astype(dtype)


def noise_like(shape, dtype=np.float16):
    """Incompressible: LZF gives up and libhdf5 stores the chunk raw with the filter-mask bit set."""
    return rng.standard_normal(shape).astype(dtype)


def write(name, libver, datasets, group='features', attrs=True):
    path = os.path.join(OUT, name)
    f = h5py.File(path, 'w', libver=libver) if libver else h5py.File(path, 'w')
    g = f.create_group(group)
    expect, meta = {}, {}
    for key, array, opts in datasets:
        d = g.create_dataset(key, array.shape, dtype=array.dtype, **opts)
        half = max(1, array.shape[0] // 2)
        d[:half] = array[:half]                          # written in slices, like the extractor's batches (:98)
        d[half:] = array[half:]
        if attrs:
            d.attrs['pca'] = np.void(pickle.dumps({'components': 3}))
            d.attrs['min'] = np.array([-1.5, -2.0, -3.25])
            d.attrs['range'] = np.array([2.0, 4.0, 6.5])
        expect[key] = array
        meta[key] = dict(shape=list(array.shape), dtype=str(array.dtype), chunks=list(d.chunks) if d.chunks else None,
                         compression=d.compression)
    f.close()
    np.savez_compressed(path + '.expected.npz', **expect)
    with open(path + '.json', 'w') as fh:
        json.dump(dict(libver=libver or 'earliest', datasets=meta, h5py=h5py.__version__, hdf5=h5py.version.hdf5_version), fh,
                  indent=1, sort_keys=True)
    print(name, os.path.getsize(path), meta)


def main():
    os.makedirs(OUT, exist_ok=True)
    lzf = dict(compression='lzf')
    # the reference's writer, two feature sets in one file (dino + lseg), compressible + incompressible chunks
    write('ref_latest.hdf', 'latest', [('dino', feature_like((6, 12, 16, 8)), lzf), ('lseg', noise_like((6, 5, 7, 32)), lzf)])
    # the same calls without libver='latest': superblock 0, symbol-table groups, v1 headers, B-tree v1 chunk index
    write('ref_earliest.hdf', None, [('dino', feature_like((6, 12, 16, 8)), lzf), ('lseg', noise_like((6, 5, 7, 32)), lzf)])
    # ragged chunk grid (edge chunks are stored padded), both index generations
    ragged = dict(compression='lzf', chunks=(3, 2, 4, 8))
    write('ragged_latest.hdf', 'latest', [('dino', feature_like((7, 5, 6, 8)), ragged)])
    write('ragged_earliest.hdf', None, [('dino', feature_like((7, 5, 6, 8)), ragged)])
    # > 1024 chunks: paged fixed-array data block (latest) / multi-level B-tree v1 (earliest)
    many = dict(compression='lzf', chunks=(1, 2, 2, 8))
    write('paged_latest.hdf', 'latest', [('dino', feature_like((66, 8, 8, 8)), many)])
    write('deep_earliest.hdf', None, [('dino', feature_like((66, 8, 8, 8)), many)])
    # one chunk (single-chunk index with the filtered size in the layout message)
    write('single_latest.hdf', 'latest', [('dino', feature_like((2, 3, 4, 8)), dict(compression='lzf', chunks=(2, 3, 4, 8)))])
    # gzip + shuffle + fletcher32, float32
    write('gzip_latest.hdf', 'latest', [('dino', feature_like((6, 12, 16, 8), np.float32),
                                         dict(compression='gzip', shuffle=True, fletcher32=True))])
    write('gzip_earliest.hdf', None, [('dino', feature_like((6, 12, 16, 8), np.float32),
                                       dict(compression='gzip', compression_opts=9, shuffle=True))])
    # no filter: contiguous, and chunked without filters (fixed array of plain addresses)
    write('plain_latest.hdf', 'latest', [('dino', noise_like((4, 6, 8, 8)), {}),
                                         ('chunked', noise_like((4, 6, 8, 8)), dict(chunks=(2, 3, 8, 8))),
                                         ('bytes', (rng.integers(0, 255, (5, 9))).astype(np.uint8), {}),
                                         ('be', feature_like((3, 4), np.dtype('>f4')), {})], attrs=False)
    write('plain_earliest.hdf', None, [('dino', noise_like((4, 6, 8, 8)), {}),
                                       ('chunked', noise_like((4, 6, 8, 8)), dict(chunks=(2, 3, 8, 8)))])


if __name__ == '__main__':
    main()
