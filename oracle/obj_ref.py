"""TEST INFRASTRUCTURE ONLY — CPU restatement of load_mesh (reference src/mesh.rs:9-73) together with
the part of the tobj crate it relies on.  Only tests/ may import this; the product loaders are
pathtracer_amd/host/mesh.hpp (C++) and pathtracer_amd/mesh.py.

PARITY UNPINNED: tobj 0.1.2 (Cargo.lock) is a crates.io dependency that is not vendored under
/root/reference, and the reference holds no OBJ fixture or loader test.  The structure below restates
tobj's published algorithm from its documentation and source as remembered: a line-oriented parser
that collects faces into the current model and closes the model at `o`/`g` and whenever `usemtl`
changes the material while faces are pending; per model, `export_faces` turns each distinct
(v, vt, vn) corner into one vertex of a single-index mesh; polygons are fanned; negative indices are
relative.  mesh.rs then builds one Triangle per index triple with the model's material_id mapped
as in mesh.rs:39-66.  Written as literally as possible (models, single-index meshes) so that it
checks the product loaders' shortcut — "each face takes the material current when it is read" —
rather than sharing it.
"""
import os

import numpy as np


def _fix(i, n):
    return n + i if i < 0 else i - 1


def _parse_corner(tok, npos, ntex, nnrm):
    parts = tok.split("/")
    v = _fix(int(parts[0]), npos)
    vt = _fix(int(parts[1]), ntex) if len(parts) > 1 and parts[1] else None
    vn = _fix(int(parts[2]), nnrm) if len(parts) > 2 and parts[2] else None
    return (v, vt, vn)


def _export_faces(pos, tex, nrm, faces, material_id):
    index_map, mesh = {}, {"positions": [], "normals": [], "texcoords": [], "indices": [], "material_id": material_id}

    def add_vertex(c):
        if c not in index_map:
            v, vt, vn = c
            index_map[c] = len(mesh["positions"]) // 3
            mesh["positions"] += pos[3 * v:3 * v + 3]
            if vt is not None:
                mesh["texcoords"] += tex[2 * vt:2 * vt + 2]
            if vn is not None:
                mesh["normals"] += nrm[3 * vn:3 * vn + 3]
        mesh["indices"].append(index_map[c])

    for f in faces:
        a, b = f[0], f[1]
        for c in f[2:]:
            add_vertex(a)
            add_vertex(b)
            add_vertex(c)
            b = c
    return mesh


def _load_mtl(path):
    mats, cur = [], None
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "newmtl":
                cur = {"name": t[1], "ambient": [0.0] * 3, "diffuse": [0.0] * 3, "shininess": 0.0, "dissolve": 1.0}
                mats.append(cur)
            elif t[0] == "Ka":
                cur["ambient"] = [float(x) for x in t[1:4]]
            elif t[0] == "Kd":
                cur["diffuse"] = [float(x) for x in t[1:4]]
            elif t[0] == "Ns":
                cur["shininess"] = float(t[1])
            elif t[0] == "d":
                cur["dissolve"] = float(t[1])
    return mats


def tobj_load_obj(path):
    pos, tex, nrm, faces = [], [], [], []
    models, materials, mat_map = [], [], {}
    name, mat_id = "unnamed_object", None
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                pos += [float(x) for x in t[1:4]]
            elif t[0] == "vt":
                tex += [float(x) for x in t[1:3]]
            elif t[0] == "vn":
                nrm += [float(x) for x in t[1:4]]
            elif t[0] == "f":
                faces.append([_parse_corner(c, len(pos) // 3, len(tex) // 2, len(nrm) // 3) for c in t[1:]])
            elif t[0] in ("o", "g"):
                if faces:
                    models.append({"name": name, "mesh": _export_faces(pos, tex, nrm, faces, mat_id)})
                    faces = []
                name = " ".join(t[1:])
            elif t[0] == "mtllib":
                for m in _load_mtl(os.path.join(os.path.dirname(path), t[1])):
                    mat_map[m["name"]] = len(materials)
                    materials.append(m)
            elif t[0] == "usemtl":
                new_mat = mat_map.get(t[1])
                if mat_id != new_mat and faces:
                    models.append({"name": name, "mesh": _export_faces(pos, tex, nrm, faces, mat_id)})
                    faces = []
                mat_id = new_mat
    if faces:
        models.append({"name": name, "mesh": _export_faces(pos, tex, nrm, faces, mat_id)})
    return models, materials


def load_mesh(path, backup_material):
    """mesh.rs:9-73 -> list of (positions 3x3, normals 3x3, material) with material =
    (kind, a, b, (r, g, b)), numbers rounded to f32."""
    f32 = lambda x: float(np.float32(x))  # noqa: E731
    models, materials = tobj_load_obj(path)
    tris = []
    for model in models:
        mesh = model["mesh"]
        positions = [mesh["positions"][i:i + 3] for i in range(0, len(mesh["positions"]), 3)]
        normals = [mesh["normals"][i:i + 3] for i in range(0, len(mesh["normals"]), 3)]
        for k in range(0, len(mesh["indices"]), 3):
            ix = mesh["indices"][k:k + 3]
            if mesh["material_id"] is not None:
                m = materials[mesh["material_id"]]
                if np.float32(m["dissolve"]) < np.float32(1.0):
                    mat = (1, 1.0, f32(1.3), tuple(f32(c) for c in m["ambient"]))
                else:
                    mat = (0, f32(m["shininess"]), 0.0, tuple(f32(c) for c in m["diffuse"]))
            else:
                mat = backup_material
            tris.append(([[f32(c) for c in positions[i]] for i in ix], [[f32(c) for c in normals[i]] for i in ix], mat))
    return tris
