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
    open square lattices) get their 2-colouring from the components of the bipartite double cover; other graphs a vectorised
    independent-set colouring (Luby rounds with random priorities, fixed seed: reproducible)."""
    n = A.shape[0]
    P = sp.csr_matrix((np.ones(A.nnz, dtype=np.int8), A.indices, A.indptr), shape=A.shape)
    P = (P + P.T).tocsr()
    P.setdiag(0)
    P.eliminate_zeros()
    color = np.full(n, -1, dtype=np.int64)
    rows = np.repeat(np.arange(n), np.diff(P.indptr))
    # --- bipartite attempt, one pass for any number of components: in the bipartite double cover (nodes (i, 0), (i, 1), an edge
    # (i, a) - (j, 1 - a) for every edge i - j) a bipartite component of the graph splits into its two colour classes, a
    # non-bipartite one stays connected.  colour(i) = [component of (i, 0) > component of (i, 1)] is then a proper 2-colouring.
    cover = sp.bmat([[None, P], [P, None]], format="csr") if P.nnz else sp.csr_matrix((2 * n, 2 * n))
    _, lab = csgraph.connected_components(cover, directed=False)
    l0, l1 = lab[:n], lab[n:]
    if not np.any((l0 == l1) & (np.diff(P.indptr) > 0)):
        two = (l0 > l1).astype(np.int64)
        two[np.diff(P.indptr) == 0] = 0          # isolated sites: colour 0
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
