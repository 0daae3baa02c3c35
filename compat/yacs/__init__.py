"""`yacs` as far as the reference's config files need it: `from yacs.config import CfgNode` (compat/README.md)."""
