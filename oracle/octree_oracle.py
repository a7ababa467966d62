"""ORACLE (test infrastructure): pure-Python restatement of RayTracer::createOctTree + loadOctTree
(reference source/Graphics/RayTracer.cpp:503-519, 523-704, 706-788), for small scenes only.

It cannot be compiled from the reference here (entt + DirectXCollision.h), so this is a reading of the
source, in float32 arithmetic with the same operation order. Entities are (id, aabb_min, aabb_max,
is_mesh) in the order the reference's entt views yield them (reverse creation order)."""
import numpy as np

F = np.float32
OFFS = np.array([[1, -1, 1], [-1, -1, 1], [1, 1, 1], [-1, 1, 1], [1, -1, -1], [-1, -1, -1], [1, 1, -1], [-1, 1, -1]], dtype=F)  # :579-590
FLT_MAX = np.finfo(F).max


def area(mn, mx):  # Utilities.h:78-82
    e = (mx - mn).astype(F)
    return F(F(e[0] * e[1]) + F(e[1] * e[2])) + F(e[2] * e[0])


def intersects(amn, amx, bmn, bmx):  # Utilities.h:84-90
    ca = ((amx + amn).astype(F) * F(0.5)).astype(F)
    cb = ((bmx + bmn).astype(F) * F(0.5)).astype(F)
    d = np.abs((cb - ca).astype(F))
    e = (((amx - amn).astype(F) * F(0.5)).astype(F) + ((bmx - bmn).astype(F) * F(0.5)).astype(F)).astype(F)
    return bool(d[0] < e[0] and d[1] < e[1] and d[2] < e[2])


def build(root_min, root_max, entities, max_depth, max_leaf_objects):
    """entities: list of dict(id, mn, mx). Returns list of nodes: dict(mn, mx, entities[ids], children[8])."""
    nodes = [dict(mn=np.asarray(root_min, F), mx=np.asarray(root_max, F), ents=list(entities), children=[None] * 8)]
    stack = [(0, 0)]
    while stack:
        idx, depth = stack.pop()
        node = nodes[idx]
        if depth >= max_depth or len(node["ents"]) <= max_leaf_objects:
            continue
        mn, mx = node["mn"].copy(), node["mx"].copy()
        centre = ((mx + mn).astype(F) * F(0.5)).astype(F)
        mx = (mx - centre).astype(F); mn = (mn - centre).astype(F)
        mn = (mn * F(0.5)).astype(F); mx = (mx * F(0.5)).astype(F)
        mx = (mx + centre).astype(F); mn = (mn + centre).astype(F)
        node_area = area(node["mn"], node["mx"])
        total = len(node["ents"])
        pool = [e for e in reversed(node["ents"]) if F(area(e["mn"], e["mx"]) / node_area) < F(0.75)]
        ext = ((mx - mn).astype(F) * F(0.5)).astype(F)
        for i in range(8):
            cmn = (mn + (OFFS[i] * ext).astype(F)).astype(F)
            cmx = (mx + (OFFS[i] * ext).astype(F)).astype(F)
            child = []
            for k in range(len(pool) - 1, -1, -1):
                if intersects(cmn, cmx, pool[k]["mn"], pool[k]["mx"]):
                    child.append(pool[k])
                    del pool[k]
            if not child or len(child) == total:
                continue
            for ce in child:
                for j, pe in enumerate(node["ents"]):
                    if pe["id"] == ce["id"]:
                        del node["ents"][j]
                        break
            bmn, bmx = np.full(3, FLT_MAX, F), np.full(3, -FLT_MAX, F)
            for e in child:  # refitOctTreeNode :503-519
                bmn = np.minimum(np.minimum(bmn, e["mn"]), e["mx"]); bmx = np.maximum(np.maximum(bmx, e["mn"]), e["mx"])
            nodes.append(dict(mn=bmn, mx=bmx, ents=child, children=[None] * 8))
            node["children"][i] = len(nodes) - 1
            stack.append((len(nodes) - 1, depth + 1))
    return nodes


def flatten(nodes):
    """loadOctTree: per node (first child, num children, instance ids in table order)."""
    out = []
    for n in nodes:
        kids = [c for c in n["children"] if c is not None]
        out.append(dict(mn=n["mn"], mx=n["mx"], first=kids[0] if kids else 0xFFFFFFFF, num=len(kids), ids=[e["id"] for e in n["ents"]]))
    return out
