"""The two workloads of the reference's examples/ directory, restated as library objects."""
