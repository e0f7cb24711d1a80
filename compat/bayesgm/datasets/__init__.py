"""bayesgm.datasets -> bayesgm_amd.datasets (the same objects)."""
from bayesgm_amd.datasets import (Base_sampler, Gaussian_sampler, Sim_Colangelo_sampler, Sim_Hirano_Imbens_sampler, Sim_Sun_sampler,
                                  simulate_z_hetero)

__all__ = ["Base_sampler", "Sim_Hirano_Imbens_sampler", "Sim_Sun_sampler", "Sim_Colangelo_sampler", "Gaussian_sampler", "simulate_z_hetero"]


def __getattr__(name):
    raise AttributeError("bayesgm.datasets.%s is outside the hot path bayesgm_amd implements (available: %s)" % (name, ", ".join(__all__)))
