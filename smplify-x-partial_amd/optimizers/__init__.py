"""Optimisers of the fitting path (mirror of smplifyx/optimizers)."""
