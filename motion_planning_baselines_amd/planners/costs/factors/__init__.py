"""Factor objects of the reference's cost layer (mp_baselines/planners/costs/factors), device-resident constants
plus HIP-served error terms.  The planners of this package fuse the same arithmetic into their kernels; these
classes exist for callers that use the factors directly."""
