"""Host-side graph preparation for the colour-parallel sparse kernel (K5): symmetric CSR and a proper colouring.

The reference has no such step: it keeps every model as a dense N x N matrix and walks the sites one by one
(tsu/gibbs.py:128-162, tsu/models/ising.py:52-97).  The colouring decides which sites may be updated together; the
result of a sweep equals the reference's sequential loop run in the colour-major visiting order."""
from typing import Tuple

import numpy as np
import scipy.sparse as sp
from scipy.sparse import csgraph


def canonical_csr(J) -> sp.csr_matrix:
    """CSR with sorted column indices and no explicit zeros (float64)."""
    A = sp.csr_matrix(J, dtype=np.float64)
    A.sum_duplicates()
    A.eliminate_zeros()
    A.sort_indices()
    return A


def color_graph(A: sp.csr_matrix, max_colors: int = 64) -> Tuple[np.ndarray, np.ndarray]:
    """Proper colouring of the coupling graph of ``A`` (its symmetrised sparsity pattern, self-loops ignored).

    Returns ``(color_offsets, order)``: ``order`` lists the sites colour by colour (ascending site number inside a
    colour), ``color_offsets[c]:color_offsets[c+1]`` delimits colour c.  Bipartite graphs (chains, trees, even rings,
    open square lattices) get their 2-colouring from a breadth-first search; other graphs a vectorised
    independent-set colouring (Luby rounds with random priorities, fixed seed: reproducible)."""
    n = A.shape[0]
    P = sp.csr_matrix((np.ones(A.nnz, dtype=np.int8), A.indices, A.indptr), shape=A.shape)
    P = (P + P.T).tocsr()
    P.setdiag(0)
    P.eliminate_zeros()
    color = np.full(n, -1, dtype=np.int64)
    # --- bipartite attempt: BFS depth parity per connected component
    n_comp, labels = csgraph.connected_components(P, directed=False)
    depth = np.zeros(n, dtype=np.int64)
    seen = np.zeros(n, dtype=bool)
    roots = np.full(n_comp, -1, dtype=np.int64)
    first = np.unique(labels, return_index=True)[1]
    roots[labels[first]] = first
    if n_comp > max(1024, n // 4):
        # many tiny components (mostly isolated sites): isolated sites need no search
        deg = np.diff(P.indptr)
        seen[deg == 0] = True
    for r in roots:
        if seen[r]:
            continue
        nodes, pred = csgraph.breadth_first_order(P, int(r), directed=False, return_predecessors=True)
        d = np.zeros(nodes.size, dtype=np.int64)
        # nodes come in BFS order: a predecessor always precedes its children
        idx = {int(v): k for k, v in enumerate(nodes)} if nodes.size < 64 else None
        if idx is not None:
            for k, v in enumerate(nodes[1:], 1):
                d[k] = d[idx[int(pred[v])]] + 1
            depth[nodes] = d
        else:
            dep = np.zeros(n, dtype=np.int64)
            for v in nodes[1:]:
                dep[v] = dep[pred[v]] + 1
            depth[nodes] = dep[nodes]
        seen[nodes] = True
    two = depth & 1
    rows = np.repeat(np.arange(n), np.diff(P.indptr))
    if not np.any(two[rows] == two[P.indices]):
        color = two
    else:
        # --- general graph: repeated maximal-independent-set extraction
        rng = np.random.default_rng(12345)
        deg = np.diff(P.indptr)
        has_nb = deg > 0
        starts = P.indptr[:-1][has_nb]            # np.maximum.reduceat over the rows that have neighbours
        left = np.ones(n, dtype=bool)
        c = 0
        while left.any():
            if c >= max_colors:
                raise ValueError(f"graph needs more than {max_colors} colours: use the dense path")
            cand = left.copy()
            chosen = np.zeros(n, dtype=bool)
            while cand.any():
                pr = np.where(cand, rng.random(n), -1.0)
                # best priority among the candidate neighbours of every site
                nb = np.full(n, -1.0)
                if starts.size:
                    nb[has_nb] = np.maximum.reduceat(pr[P.indices], starts)
                win = cand & (pr > nb)
                chosen |= win
                # winners and their neighbours leave the candidate set
                hit = np.zeros(n, dtype=bool)
                hit[P.indices[win[rows]]] = True
                cand &= ~win & ~hit
            color[chosen] = c
            left &= ~chosen
            c += 1
    n_colors = int(color.max()) + 1
    order = np.argsort(color, kind="stable").astype(np.int32)
    offsets = np.zeros(n_colors + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(np.bincount(color, minlength=n_colors))
    return offsets, order
