from stmask_amd.dcn_v2 import DCN, DCNv2  # noqa: F401
