"""load_mesh of the reference (src/mesh.rs:9-73) for the Python host side: a minimal Wavefront
OBJ / MTL reader producing the arrays PathTracer.set_triangles takes.  Same rules as
pathtracer_amd/host/mesh.hpp (the C++ host mirror), which documents what is taken from tobj — and
that parity with the tobj crate itself (in particular the order in which its models, and so the
triangles, come out: mesh.rs:10-29) is UNPINNED: the crate is absent and the reference holds no
OBJ fixture.
Numbers go through Python's float() and are then rounded to f32; for inputs of up to 17
significant digits that equals a direct correctly rounded f32 parse except on exact ties."""
import os

import numpy as np

KIND_DIFFUSE, KIND_DIELECTRIC, KIND_EMISSIVE = 0, 1, 2


def _load_mtl(path):
    mats, by_name, cur = [], {}, None
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "newmtl" and len(t) >= 2:
                cur = {"Ka": (0.0, 0.0, 0.0), "Kd": (0.0, 0.0, 0.0), "Ns": 0.0, "d": 1.0}
                by_name[t[1]] = len(mats)
                mats.append(cur)
            elif cur is not None and t[0] in ("Ka", "Kd") and len(t) >= 4:
                cur[t[0]] = tuple(float(x) for x in t[1:4])
            elif cur is not None and t[0] in ("Ns", "d") and len(t) >= 2:
                cur[t[0]] = float(t[1])
    return mats, by_name


def _corner(tok, n_pos, n_nrm):
    parts = tok.split("/")
    if not parts[0]:
        raise ValueError("face corner without a position")
    v = int(parts[0])
    vi = n_pos + v if v < 0 else v - 1
    if len(parts) < 3 or not parts[2]:
        raise ValueError("face corner without a normal (the reference indexes normals unconditionally)")
    n = int(parts[2])
    ni = n_nrm + n if n < 0 else n - 1
    if not (0 <= vi < n_pos) or not (0 <= ni < n_nrm):
        raise ValueError("face index out of range")
    return vi, ni


def load_mesh(path, backup_material):
    """-> (positions N x 9, normals N x 9, material_id N, material rows) in file order; the rows are
    (kind, a, b, (r, g, b)) tuples as scenes.wrap_mesh takes them.
    backup_material: (kind, a, b, (r, g, b)) used for faces without a material (mesh.rs:55-66)."""
    positions, normals, tris_v, tris_n, tri_mat = [], [], [], [], []
    mats_out, mat_index = [], {}
    mtl, mtl_by_name, material_id = [], {}, None

    def material_slot(key, value):
        if key not in mat_index:
            mat_index[key] = len(mats_out)
            mats_out.append(value)
        return mat_index[key]

    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] in ("v", "vn") and len(t) >= 4:
                (positions if t[0] == "v" else normals).append([float(x) for x in t[1:4]])
            elif t[0] == "mtllib" and len(t) >= 2:
                m, by = _load_mtl(os.path.join(os.path.dirname(path), t[1]))
                for name, i in by.items():
                    mtl_by_name[name] = len(mtl) + i
                mtl += m
            elif t[0] == "usemtl" and len(t) >= 2:
                material_id = mtl_by_name.get(t[1])
            elif t[0] == "f":
                if len(t) < 4:
                    raise ValueError("face with fewer than 3 corners")
                corners = [_corner(c, len(positions), len(normals)) for c in t[1:]]
                if material_id is None:
                    slot = material_slot("backup", backup_material)
                else:
                    e = mtl[material_id]
                    if np.float32(e["d"]) < np.float32(1.0):
                        slot = material_slot(("m", material_id), (KIND_DIELECTRIC, 1.0, 1.3, e["Ka"]))
                    else:
                        slot = material_slot(("m", material_id), (KIND_DIFFUSE, e["Ns"], 0.0, e["Kd"]))
                for k in range(1, len(corners) - 1):
                    tris_v.append((corners[0][0], corners[k][0], corners[k + 1][0]))
                    tris_n.append((corners[0][1], corners[k][1], corners[k + 1][1]))
                    tri_mat.append(slot)
    P = np.asarray(positions, np.float64).astype(np.float32).reshape(-1, 3)
    Nn = np.asarray(normals, np.float64).astype(np.float32).reshape(-1, 3)
    tv = np.asarray(tris_v, np.int64).reshape(-1, 3)
    tn = np.asarray(tris_n, np.int64).reshape(-1, 3)
    pos = P[tv].reshape(-1, 9)
    nrm = Nn[tn].reshape(-1, 9)
    return pos, nrm, np.asarray(tri_mat, np.uint32), mats_out
