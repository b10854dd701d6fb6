"""Repo-root pytest configuration: `pytest` (bare, from the root) collects tests/ only.  tools/ holds timing scripts and archived experiments that drive
entry points the library no longer exports, gpurun_out/ is scratch from GPU runs, oracle/ is the checker the tests import."""
collect_ignore = ['tools', 'gpurun_out', 'oracle', 'profiles', 'normalisr_amd', 'normalisr', 'bin', 'include']
