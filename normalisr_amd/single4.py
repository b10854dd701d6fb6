"""single=4: competition-aware association (other groupings as covariates), association.py:421-576,926-980."""


def association_tests_single4(dx, dy, dc, lowmem=True, return_dot=True, return_stats=False, **ka):
	raise NotImplementedError('single=4 device path not built yet')
