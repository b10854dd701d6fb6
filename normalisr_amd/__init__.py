"""normalisr_amd: MI355X-native implementation of Normalisr's linear-association hot path
(norm.de / norm.coex).  Importing submodules mirrors the reference package layout:
normalisr_amd.normalisr, .de, .coex, .association, .parallel, .run."""
__all__ = ['association', 'binnet', 'coex', 'de', 'norm', 'normalisr', 'parallel', 'run']
__version__ = '0.1.0'
