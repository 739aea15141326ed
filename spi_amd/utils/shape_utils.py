"""Shape export of the density grid (drop-in surface of eg3d/shape_utils.py:40-104 as spi/utils/video_utils.py:212-218 uses it).

``convert_sdf_samples_to_ply(grid, origin, voxel_size, path, offset=None, scale=None, level=0.0)`` extracts the ``level``
iso-surface of a 3-D scalar grid and writes a binary little-endian ``.ply`` (vertex x/y/z float32, face ``vertex_indices`` int32
lists -- the layout ``plyfile`` writes for the reference); ``write_mrc`` / ``read_mrc`` / ``convert_mrc`` cover the ``.mrc``
(MRC2014, mode 2 = float32) side.

The reference delegates the surface extraction to ``skimage.measure.marching_cubes`` and the files to ``plyfile`` / ``mrcfile``:
third-party code that is neither under /root/reference nor installed here ("parity unpinned" at that edge).
The extraction here is a table-driven MARCHING CUBES (round 3; rounds 1-2 had marching tetrahedra, whose vertices also sit on the
cells' diagonals and are therefore not comparable with the reference's mesh): one vertex per grid edge crossing the level, linearly
interpolated -- the vertex set every marching-cubes variant, skimage's included, produces (its Lewiner variant adds a cell-centre
vertex in a few ambiguous configurations) -- 256-case table with consistently resolved face ambiguities, closed and consistently
oriented (tests/test_host_cpu.py: analytic sphere, a two-blob field with ambiguous faces, the vertex-count identity).  Host-side
numpy: post-processing, not the hot path.
"""
import os
import struct

import numpy as np

# ---- marching cubes tables, generated at import (no 256-row literal to mistype) -----------------------------------------------------
# Cube corners in the classic numbering (Lorensen & Cline 1987 / Bourke's public tables): bottom face 0-1-2-3 counter-clockwise seen
# from above, top face 4-5-6-7 over it; the 12 edges in the classic order.
_CORNERS = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.int64)
_EDGES = np.array([[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4], [0, 4], [1, 5], [2, 6], [3, 7]], dtype=np.int64)
# the six faces, corners counter-clockwise as seen from OUTSIDE the cube
_FACES = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
_EDGE_ID = {(int(a_), int(b_)): i for i, (a_, b_) in enumerate(_EDGES)}
_EDGE_ID.update({(b_, a_): i for (a_, b_), i in list(_EDGE_ID.items())})


def _on_common_face(ea, eb):
    ca, cb = set(_EDGES[ea].tolist()), set(_EDGES[eb].tolist())
    return any(ca <= set(f) and cb <= set(f) for f in _FACES)


def _triangulate(loop):
    """Triangles (same orientation as the loop) of a closed loop of cut edges such that no DIAGONAL joins two cut edges of one cube face:
    such a diagonal lies in that face, the neighbouring cell may draw the same one, and the mesh edge would then carry four triangles."""
    n = len(loop)
    if n == 3:
        return [tuple(loop)]
    for k in range(1, n - 1):                                    # triangle (loop[0], loop[k], loop[n-1]) + the two sub-polygons on its sides
        if (k > 1 and _on_common_face(loop[0], loop[k])) or (k < n - 2 and _on_common_face(loop[k], loop[n - 1])):
            continue
        left = _triangulate(loop[:k + 1]) if k > 1 else []
        right = _triangulate(loop[k:]) if k < n - 2 else []
        if left is not None and right is not None:
            return left + [(loop[0], loop[k], loop[n - 1])] + right
    return None


def _build_mc_table():
    """table[mask] = triangles (triples of edge ids) of the iso-surface in a cell whose corner i is inside (value >= level) iff bit i of
    mask is set.  Per face the cut edges are joined around every run of inside corners (an ambiguous face -- two diagonal inside corners
    -- separates them: the choice depends on the face's four flags only, so both cells sharing a face draw the same segments and the
    surface is watertight, unlike the original 1987 table); segments run from the edge where a counter-clockwise walk along the face
    boundary ENTERS the inside run to the edge where it leaves it, which orients every loop with its normal towards the outside
    (lower values); the closed loops are triangulated without diagonals inside a cube face (_triangulate).  Unambiguous cases are the
    classic ones up to the choice of diagonals."""
    table = []
    for mask in range(256):
        inside = [(mask >> i) & 1 for i in range(8)]
        nxt = {}
        for f in _FACES:
            flags = [inside[c] for c in f]
            if sum(flags) in (0, 4):
                continue
            for i in range(4):
                if flags[i] and not flags[i - 1]:                # a run of inside corners starts at corner i: entered through edge (i-1, i)
                    j = i
                    while flags[(j + 1) % 4]:
                        j += 1
                    enter = _EDGE_ID[(f[i - 1], f[i])]
                    leave = _EDGE_ID[(f[j % 4], f[(j + 1) % 4])]
                    assert enter not in nxt
                    nxt[enter] = leave
        tris, seen = [], set()
        for e0 in sorted(nxt):
            if e0 in seen:
                continue
            loop, e = [], e0
            while e not in seen:
                seen.add(e)
                loop.append(e)
                e = nxt[e]
            assert e == e0 and len(loop) >= 3
            t_ = _triangulate(loop)
            assert t_ is not None, (mask, loop)
            tris += t_
        assert len(seen) == sum(1 for a_, b_ in _EDGES if inside[a_] != inside[b_])          # every cut edge carries exactly one vertex
        table.append(tris)
    width = max(len(t) for t in table)
    arr = np.full((256, width, 3), -1, dtype=np.int64)
    for m, t in enumerate(table):
        if t:
            arr[m, :len(t)] = t
    return arr


_MC_TABLE = _build_mc_table()                                    # [256, T, 3] edge ids, -1 padded


def marching_cubes(vol, level=0.0, spacing=(1.0, 1.0, 1.0), slab=32):
    """Table-driven marching cubes: -> (verts float64 [V,3] in index units * spacing, faces int64 [F,3]).
    One vertex per grid edge that crosses ``level`` (linear interpolation along the edge, what skimage.measure.marching_cubes -- the
    reference's extractor, eg3d/shape_utils.py:60-62 -- places there too), shared between the cells around the edge; triangles are
    oriented with their normal pointing from values >= level towards values < level (skimage's default gradient_direction='descent').
    Works slab by slab over the first axis, touching only the cut cells."""
    vol = np.asarray(vol)
    assert vol.ndim == 3 and min(vol.shape) >= 2
    nx, ny, nz = vol.shape
    nn = nx * ny * nz
    tri_keys = []                                                # [T, 3, 2] grid-node pairs of the three cut edges
    for x0 in range(0, nx - 1, slab):
        x1 = min(x0 + slab, nx - 1)
        sub = vol[x0:x1 + 1]
        c = [sub[dx:sub.shape[0] - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz] for dx, dy, dz in _CORNERS]
        mask = np.zeros(c[0].shape, dtype=np.int64)
        for i, cc in enumerate(c):
            mask |= (cc >= level).astype(np.int64) << i
        ii, jj, kk = np.nonzero((mask != 0) & (mask != 255))
        if ii.size == 0:
            continue
        nodes = np.stack([((ii + x0 + dx) * ny + (jj + dy)) * nz + (kk + dz) for dx, dy, dz in _CORNERS], axis=1)        # [A, 8]
        tris = _MC_TABLE[mask[ii, jj, kk]]                       # [A, T, 3] edge ids
        cell, slot = np.nonzero(tris[:, :, 0] >= 0)
        ends = _EDGES[tris[cell, slot]]                          # [n, 3, 2] corner ids
        tri_keys.append(np.take_along_axis(nodes[cell][:, None, :], ends.reshape(len(cell), 1, 6), axis=2).reshape(-1, 3, 2))
    if not tri_keys:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    keys = np.sort(np.concatenate(tri_keys), axis=-1)            # an edge = its (smaller, larger) node
    flat = keys.reshape(-1, 2)
    uniq, inv = np.unique(flat[:, 0] * nn + flat[:, 1], return_inverse=True)
    a, b = uniq // nn, uniq % nn

    def pos(n):
        return np.stack([n // (ny * nz), (n // nz) % ny, n % nz], axis=1).astype(np.float64)
    va, vb = vol.reshape(-1)[a].astype(np.float64), vol.reshape(-1)[b].astype(np.float64)
    t = np.where(vb != va, (level - va) / np.where(vb != va, vb - va, 1.0), 0.5)[:, None]
    verts = (pos(a) * (1 - t) + pos(b) * t) * np.asarray(spacing, dtype=np.float64)
    faces = inv.reshape(-1, 3).astype(np.int64)
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
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
    verts, faces = marching_cubes(np.asarray(numpy_3d_sdf_tensor), level=level, spacing=[voxel_size] * 3)
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
