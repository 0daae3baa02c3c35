from softmac_amd.config.cfgnode import CfgNode  # noqa: F401
