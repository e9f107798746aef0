from stmask_amd.spatial_correlation_sampler import SpatialCorrelationSampler, spatial_correlation_sample  # noqa: F401
