"""Gene-gene co-expression (mirror of the reference's coex module)."""
from .association import association_tests


def coex(dt, dc, **ka):
	"""Co-expression of all gene pairs of dt (n_gene, n_cell) given covariates dc (n_cov, n_cell).
	Same contract as reference coex.py:4-48: returns (P-values (n_gene,n_gene), dot (n_gene,n_gene),
	var (n_gene,)); dot is the covariance of the residualised genes, Pearson R = dot/sqrt(var_i var_j);
	diagonals of P-values and dot are 0.  Keyword arguments: bs (accepted), nth, dimreduce."""
	ans = association_tests(dt, None, dc, **ka)
	return (ans[0], ans[1], ans[4])


assert __name__ != "__main__"
