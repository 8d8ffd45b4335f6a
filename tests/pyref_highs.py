"""TEST INFRASTRUCTURE. The reference's integer programme for one (genotype, attempt), as `HighsSolver::define_model` states it
(/root/reference/src/solvers/highs.rs:38-100), handed to the solver the reference hands it to: HiGHS, here through scipy.optimize.milp
(scipy bundles HiGHS; the reference binds it through the `highs` crate). Options as highs.rs:103-110 leaves them — presolve on, the
default relative gap of 1e-4, one thread — and the answer decoded as highs.rs:121-131 decodes it: per read the location with the largest
column value.

The model is built from the ORACLE's GenotypeAlignments (tests/oracle_ffi.py: read_ixs / ln_prob / windows after apply_tweak, the window
distributions), so a test holds three things against each other: HiGHS on the reference's model, the oracle's bookkeeping of a read
assignment's likelihood, and the device library's exact solver (lcty_exact.cpp)."""
import numpy as np
from scipy import sparse
from scipy.optimize import Bounds, LinearConstraint, milp


def define_model(read_ixs, ln_prob, windows, gc, weight, depth_ln_prob, aln_contrib, depth_contrib):
    """highs.rs:38-100. `depth_ln_prob(w, depth)` = WindowDistr::ln_prob of window w (weight x table entry). Returns
    (c, A, lo, hi, n_read_cols, cols_of_read) for `maximise c x`."""
    n_reads = len(read_ixs) - 1
    total_windows = len(gc)
    trivial = np.zeros(total_windows, dtype=np.int64)
    nontrivial = np.zeros(total_windows, dtype=np.int64)
    depth_rows = [[] for _ in range(total_windows)]            # (column, coefficient)
    c = []
    rows_i, rows_j, rows_v, lo, hi = [], [], [], [], []
    n_rows = 0
    cols_of_read = []
    for rp in range(n_reads):
        a, b = int(read_ixs[rp]), int(read_ixs[rp + 1])
        if b - a == 1:
            for w in windows[a]:
                trivial[int(w)] += 1
            cols_of_read.append(None)
            continue
        first = len(c)
        for t in range(a, b):
            col = len(c)
            c.append(aln_contrib * float(ln_prob[t]))
            w0, w1 = int(windows[t][0]), int(windows[t][1])
            inc = 2 if w0 == w1 else 1
            for w in (w0, w1):
                nontrivial[w] += inc
                depth_rows[w].append((col, float(inc)))
                if inc == 2:
                    break
            rows_i.append(n_rows); rows_j.append(col); rows_v.append(1.0)
        lo.append(1.0); hi.append(1.0); n_rows += 1
        cols_of_read.append((first, len(c)))
    n_read_cols = len(c)
    for w in range(total_windows):
        if weight[w] == 0.0:                                   # WindowDistr::is_trivial
            continue
        if nontrivial[w] == 0:
            continue
        row0, row1 = n_rows, n_rows + 1
        for col, coef in depth_rows[w]:
            rows_i.append(row0); rows_j.append(col); rows_v.append(coef)
        for inc in range(int(nontrivial[w]) + 1):
            col = len(c)
            c.append(depth_contrib * depth_ln_prob(w, int(trivial[w]) + inc))
            rows_i.append(row1); rows_j.append(col); rows_v.append(1.0)
            if inc > 0:
                rows_i.append(row0); rows_j.append(col); rows_v.append(-float(inc))
        lo += [0.0, 1.0]; hi += [0.0, 1.0]; n_rows += 2
    A = sparse.csr_matrix((rows_v, (rows_i, rows_j)), shape=(n_rows, len(c)))
    return np.array(c), A, np.array(lo), np.array(hi), n_read_cols, cols_of_read


def solve(read_ixs, ln_prob, windows, gc, weight, depth_ln_prob, aln_contrib, depth_contrib, time_limit=600.0, mip_rel_gap=None):
    """HighsSolver::solve_nontrivial (highs.rs:103-134). Returns (status_is_optimal, assignment[n_reads] uint16, objective, info)."""
    c, A, lo, hi, n_read_cols, cols_of_read = define_model(read_ixs, ln_prob, windows, gc, weight, depth_ln_prob, aln_contrib, depth_contrib)
    options = {"presolve": True, "time_limit": time_limit, "disp": False}
    if mip_rel_gap is not None:
        options["mip_rel_gap"] = mip_rel_gap
    res = milp(c=-c, constraints=LinearConstraint(A, lo, hi), integrality=np.ones(len(c)), bounds=Bounds(0.0, 1.0), options=options)
    n_reads = len(read_ixs) - 1
    assgn = np.zeros(n_reads, dtype=np.uint16)
    if res.x is not None:
        for rp, cols in enumerate(cols_of_read):
            if cols is not None:
                assgn[rp] = int(np.argmax(res.x[cols[0]:cols[1]]))      # F64Ext::argmax: the first of equal values
    info = {"status": int(res.status), "message": str(res.message), "columns": int(len(c)), "rows": int(A.shape[0]), "read_columns": int(n_read_cols),
            "mip_gap": getattr(res, "mip_gap", None), "mip_dual_bound": getattr(res, "mip_dual_bound", None),
            "mip_node_count": getattr(res, "mip_node_count", None)}
    return res.status == 0, assgn, (-float(res.fun) if res.fun is not None else float("nan")), info
