"""Drop-in package name of the reference (Akomand/CausalDiffAE `improved_diffusion`), backed by causaldiffae_amd."""
