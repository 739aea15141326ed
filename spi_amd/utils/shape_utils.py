"""Shape export of the density grid (drop-in surface of eg3d/shape_utils.py:40-104 as spi/utils/video_utils.py:212-218 uses it).

``convert_sdf_samples_to_ply(grid, origin, voxel_size, path, offset=None, scale=None, level=0.0)`` extracts the ``level``
iso-surface of a 3-D scalar grid and writes a binary little-endian ``.ply`` (vertex x/y/z float32, face ``vertex_indices`` int32
lists -- the layout ``plyfile`` writes for the reference); ``write_mrc`` / ``read_mrc`` / ``convert_mrc`` cover the ``.mrc``
(MRC2014, mode 2 = float32) side.

The reference delegates the surface extraction to ``skimage.measure.marching_cubes`` and the files to ``plyfile`` / ``mrcfile``:
third-party code that is neither under /root/reference nor installed here ("parity unpinned" at that edge).
The extraction here is marching TETRAHEDRA (each cell split into six tetrahedra around its main diagonal, linear
interpolation on the cut edges, vertices shared through their grid-edge key): the same iso-surface up to the triangulation
inside a cell, closed and consistently oriented (tests/test_host_cpu.py checks it on an analytic sphere).  Host-side numpy:
post-processing, not the hot path.
"""
import os
import struct

import numpy as np

_CORNERS = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.int64)
_TETS = np.array([[0, 5, 1, 6], [0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6]], dtype=np.int64)
# per inside-mask (bit i = corner i of the tetrahedron is >= level): triangles as triples of edges (a, b) between tet corners
_E = {(0, 1): 0, (1, 2): 1, (0, 2): 2, (0, 3): 3, (1, 3): 4, (2, 3): 5}
_EDGE_ENDS = np.array([[0, 1], [1, 2], [0, 2], [0, 3], [1, 3], [2, 3]], dtype=np.int64)


def _e(a, b):
    return _E[(min(a, b), max(a, b))]


_CASES = {
    0x1: [(_e(0, 1), _e(0, 2), _e(0, 3))],
    0x2: [(_e(1, 0), _e(1, 3), _e(1, 2))],
    0x4: [(_e(2, 0), _e(2, 1), _e(2, 3))],
    0x8: [(_e(3, 0), _e(3, 2), _e(3, 1))],
    0x3: [(_e(0, 3), _e(0, 2), _e(1, 3)), (_e(1, 3), _e(0, 2), _e(1, 2))],
    0x5: [(_e(0, 1), _e(2, 3), _e(0, 3)), (_e(0, 1), _e(1, 2), _e(2, 3))],
    0x6: [(_e(0, 1), _e(1, 3), _e(2, 3)), (_e(0, 1), _e(2, 3), _e(0, 2))],
}
for _m in (0x1, 0x2, 0x4, 0x8, 0x3, 0x5, 0x6):          # complementary masks cut the same edges
    _CASES[0xF ^ _m] = _CASES[_m]


def marching_tetrahedra(vol, level=0.0, spacing=(1.0, 1.0, 1.0), slab=32):
    """-> (verts float64 [V,3] in index units * spacing, faces int64 [F,3]).  Triangles are oriented with their normal pointing
    from values >= level towards values < level.  Works slab by slab over the first axis, touching only the cut cells."""
    vol = np.asarray(vol)
    assert vol.ndim == 3 and min(vol.shape) >= 2
    nx, ny, nz = vol.shape
    node_id = lambda i, j, k: (i * ny + j) * nz + k
    tri_keys = []                  # [T, 3, 2] grid-node pairs of the three cut edges
    tri_in = []                    # [T] a node on the inside (>= level) of the tetrahedron, for the orientation
    for x0 in range(0, nx - 1, slab):
        x1 = min(x0 + slab, nx - 1)
        sub = vol[x0:x1 + 1]
        c = [sub[dx:sub.shape[0] - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz] for dx, dy, dz in _CORNERS]
        lo, hi = np.minimum.reduce(c), np.maximum.reduce(c)
        ii, jj, kk = np.nonzero((lo < level) & (hi >= level))
        if ii.size == 0:
            continue
        vals = np.stack([cc[ii, jj, kk] for cc in c], axis=1).astype(np.float64)                 # [A, 8]
        nodes = np.stack([node_id(ii + x0 + dx, jj + dy, kk + dz) for dx, dy, dz in _CORNERS], axis=1)     # [A, 8]
        for tet in _TETS:
            tv, tn = vals[:, tet], nodes[:, tet]
            inside = tv >= level
            mask = inside[:, 0] * 1 + inside[:, 1] * 2 + inside[:, 2] * 4 + inside[:, 3] * 8
            for m, tris in _CASES.items():
                sel = np.nonzero(mask == m)[0]
                if sel.size == 0:
                    continue
                first_in = int(np.log2(m & -m))                                                    # lowest set bit: an inside corner
                for tri in tris:
                    ends = _EDGE_ENDS[list(tri)]                                                   # [3, 2] tet-corner indices
                    tri_keys.append(np.stack([tn[sel][:, ends[:, 0]], tn[sel][:, ends[:, 1]]], axis=-1))
                    tri_in.append(tn[sel, first_in])
    if not tri_keys:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    keys = np.concatenate(tri_keys)                                                               # [T, 3, 2]
    inside_node = np.concatenate(tri_in)
    keys = np.sort(keys, axis=-1)
    flat = keys.reshape(-1, 2)
    uniq, inv = np.unique(flat[:, 0] * (nx * ny * nz) + flat[:, 1], return_inverse=True)
    a, b = uniq // (nx * ny * nz), uniq % (nx * ny * nz)

    def pos(n):
        return np.stack([n // (ny * nz), (n // nz) % ny, n % nz], axis=1).astype(np.float64)
    va, vb = vol.reshape(-1)[a].astype(np.float64), vol.reshape(-1)[b].astype(np.float64)
    t = np.where(vb != va, (level - va) / np.where(vb != va, vb - va, 1.0), 0.5)[:, None]
    verts = (pos(a) * (1 - t) + pos(b) * t) * np.asarray(spacing, dtype=np.float64)
    faces = inv.reshape(-1, 3).astype(np.int64)
    p = verts[faces]
    nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    flip = np.einsum('ij,ij->i', nrm, p.mean(1) - pos(inside_node) * np.asarray(spacing, dtype=np.float64)) < 0
    faces[flip] = faces[flip][:, ::-1]
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])          # a value exactly on a node
    return verts, faces[keep]


def write_ply(path, verts, faces):
    """binary_little_endian 1.0: `element vertex` (x, y, z float32), `element face` (`property list uchar int vertex_indices`)."""
    verts = np.asarray(verts, dtype='<f4').reshape(-1, 3)
    faces = np.asarray(faces, dtype='<i4').reshape(-1, 3)
    hdr = ('ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n'
           'element face %d\nproperty list uchar int vertex_indices\nend_header\n' % (len(verts), len(faces)))
    rec = np.empty(len(faces), dtype=[('n', 'u1'), ('v', '<i4', (3,))])
    rec['n'], rec['v'] = 3, faces
    with open(path, 'wb') as f:
        f.write(hdr.encode('ascii'))
        f.write(verts.tobytes())
        f.write(rec.tobytes())


def read_ply(path):
    with open(path, 'rb') as f:
        raw = f.read()
    end = raw.index(b'end_header\n') + len(b'end_header\n')
    hdr = raw[:end].decode('ascii').split('\n')
    nv = int([l for l in hdr if l.startswith('element vertex')][0].split()[-1])
    nf = int([l for l in hdr if l.startswith('element face')][0].split()[-1])
    verts = np.frombuffer(raw, dtype='<f4', count=nv * 3, offset=end).reshape(nv, 3)
    rec = np.frombuffer(raw, dtype=[('n', 'u1'), ('v', '<i4', (3,))], count=nf, offset=end + nv * 12)
    return verts.copy(), rec['v'].copy()


def convert_sdf_samples_to_ply(numpy_3d_sdf_tensor, voxel_grid_origin, voxel_size, ply_filename_out, offset=None, scale=None, level=0.0):
    """eg3d/shape_utils.py:40-100: iso-surface at ``level`` with spacing ``voxel_size``, shifted by the grid origin, then
    ``/ scale`` and ``- offset`` when given."""
    verts, faces = marching_tetrahedra(np.asarray(numpy_3d_sdf_tensor), level=level, spacing=[voxel_size] * 3)
    pts = verts + np.asarray(voxel_grid_origin, dtype=np.float64).reshape(1, 3)
    if scale is not None:
        pts = pts / scale
    if offset is not None:
        pts = pts - offset
    write_ply(ply_filename_out, pts, faces)
    return pts, faces


def write_mrc(path, data):
    """MRC2014 volume, mode 2 (float32), what `mrcfile.new_mmap(path, shape=data.shape, mrc_mode=2)` + `mrc.data[:] = data`
    produces (video_utils.py:216-217): 1024-byte header, then the samples with x fastest (numpy [z, y, x] order)."""
    data = np.ascontiguousarray(data, dtype='<f4')
    assert data.ndim == 3
    nz, ny, nx = data.shape
    h = bytearray(1024)
    struct.pack_into('<3i', h, 0, nx, ny, nz)
    struct.pack_into('<i', h, 12, 2)                                   # mode 2: float32
    struct.pack_into('<3i', h, 28, nx, ny, nz)                         # mx, my, mz
    struct.pack_into('<3f', h, 40, float(nx), float(ny), float(nz))    # cell dimensions (1 A voxels)
    struct.pack_into('<3f', h, 52, 90.0, 90.0, 90.0)
    struct.pack_into('<3i', h, 64, 1, 2, 3)                            # mapc, mapr, maps
    struct.pack_into('<3f', h, 76, float(data.min()), float(data.max()), float(data.mean()))
    struct.pack_into('<i', h, 88, 1)                                   # ispg 1: volume
    h[104:108] = b'\x00\x00\x00\x00'
    struct.pack_into('<i', h, 108, 20140)                              # nversion
    h[208:212] = b'MAP '
    h[212:216] = b'\x44\x44\x00\x00'                                   # little-endian machine stamp
    struct.pack_into('<f', h, 216, float(data.std()))
    struct.pack_into('<i', h, 220, 1)
    lab = b'Created by spi_amd.utils.shape_utils'
    h[224:224 + len(lab)] = lab
    with open(path, 'wb') as f:
        f.write(bytes(h))
        f.write(data.tobytes())


def read_mrc(path):
    with open(path, 'rb') as f:
        raw = f.read()
    nx, ny, nz, mode = struct.unpack_from('<4i', raw, 0)
    if raw[208:212] != b'MAP ' or mode != 2:
        raise ValueError(f'{path}: not a little-endian MRC2014 float32 volume')
    next_ = struct.unpack_from('<i', raw, 92)[0]
    return np.frombuffer(raw, dtype='<f4', count=nx * ny * nz, offset=1024 + next_).reshape(nz, ny, nx).copy()


def convert_mrc(input_filename, output_filename, isosurface_level=1):
    """eg3d/shape_utils.py:102-104"""
    return convert_sdf_samples_to_ply(np.transpose(read_mrc(input_filename), (2, 1, 0)), [0, 0, 0], 1, output_filename, level=isosurface_level)


if __name__ == '__main__':
    import argparse
    import glob
    ap = argparse.ArgumentParser()
    ap.add_argument('input_mrc_path')
    ap.add_argument('--level', type=float, default=10, help='The isosurface level for marching cubes')
    args = ap.parse_args()
    paths = [args.input_mrc_path] if os.path.isfile(args.input_mrc_path) else glob.glob(os.path.join(args.input_mrc_path, '*.mrc'))
    for mrc_path in paths:
        convert_mrc(mrc_path, mrc_path.split('.mrc')[0] + '.ply', isosurface_level=args.level)
