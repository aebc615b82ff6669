"""Host-side skeleton topology (init only): edge graph, neighbourhoods, pooling hierarchy.

Mirrors the *behaviour* of the reference helpers (reference paths):
  get_edges          skeleton.py:306-315
  calc_edge_mat      skeleton.py:364-387   (here: breadth-first search on the edge-adjacency graph)
  find_neighbor      skeleton.py:390-410
  SkeletonPool.__init__ derivation of seq_list / pooling_list / new_edges   skeleton.py:173-207
"""
import json
import os
from collections import deque

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
SMPL24_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)


def resolve_asset(path):
    """The reference opens cwd-relative ./utils/data/...; accept that, else fall back to the packaged copy."""
    if path and os.path.exists(path):
        return path
    cand = os.path.join(ASSETS, os.path.basename(path))
    if os.path.exists(cand):
        return cand
    raise FileNotFoundError(path)


def edges_from_parents(parent_list):
    n = len(parent_list)
    # the virtual root edge (0, n) comes first so that pooling always keeps it
    return [(0, n)] + [(int(parent_list[j]), j) for j in range(1, n)]


def get_edges(parent_json):
    with open(resolve_asset(parent_json), "r") as f:
        return edges_from_parents(json.load(f))


def calc_edge_mat(edges):
    """All-pairs hop distance between edges; two edges touch when they share a joint id."""
    m = len(edges)
    touch = [[] for _ in range(m)]
    by_joint = {}
    for i, e in enumerate(edges):
        for j in (e[0], e[1]):
            by_joint.setdefault(j, []).append(i)
    for ids in by_joint.values():
        for a in ids:
            for b in ids:
                if a != b and b not in touch[a]:
                    touch[a].append(b)
    big = 100000
    mat = [[big] * m for _ in range(m)]
    for s in range(m):
        mat[s][s] = 0
        dq = deque([s])
        while dq:
            u = dq.popleft()
            for v in touch[u]:
                if mat[s][v] == big:
                    mat[s][v] = mat[s][u] + 1
                    dq.append(v)
    return mat


def find_neighbor(edges, d):
    mat = calc_edge_mat(edges)
    return [[j for j in range(len(edges)) if mat[i][j] <= d] for i in range(len(edges))]


def derive_pooling(edges, last_pool=False):
    """Returns (seq_list, pooling_list, new_edges) exactly as SkeletonPool derives them."""
    deg = {}
    for a, b in edges:
        deg[a] = deg.get(a, 0) + 1
        deg[b] = deg.get(b, 0) + 1
    seqs = []
    # explicit stack instead of recursion; children are visited in edge order
    stack = [(0, [])]
    while stack:
        joint, seq = stack.pop()
        if deg[joint] > 2 and joint != 0:
            seqs.append(seq)
            seq = []
        if deg[joint] == 1:
            seqs.append(seq)
            continue
        kids = [(e[1], seq + [idx]) for idx, e in enumerate(edges) if e[0] == joint]
        stack.extend(reversed(kids))
    pooling, new_edges = [], []
    for seq in seqs:
        if last_pool:
            pooling.append(list(seq))
            continue
        s = list(seq)
        if len(s) & 1:
            pooling.append([s[0]])
            new_edges.append(edges[s[0]])
            s = s[1:]
        for k in range(0, len(s), 2):
            pooling.append([s[k], s[k + 1]])
            new_edges.append([edges[s[k]][0], edges[s[k + 1]][1]])
    return seqs, pooling, new_edges


def stride_schedule(train_seq_len, num_layers):
    """Per-level conv stride and the time-step ladder (seq_two_hier_sa_vae.py:79-90, 107-118)."""
    strides = []
    for i in range(num_layers):
        if train_seq_len == 8:
            strides.append(1 if i in (0, num_layers - 1) else 2)
        elif train_seq_len == 16:
            strides.append(1 if i == 0 else 2)
        else:
            strides.append(2)
    steps = [train_seq_len]
    for s in strides:
        steps.append(steps[-1] // s)
    return strides, steps
