"""Data scalers and the self-conditioning hook with the reference's factory surface (``utils.py:33-150``).

The sampling hot path uses the inverse scaler inside ``ds_post_process`` (HIP); these closures exist so code
written against the reference (``get_data_inverse_scaler(config)``, ``get_self_cond_fn(config)``) keeps working
and so the sampler can verify that the configuration matches what the kernel hard-codes.
"""
from __future__ import annotations


def _factors(config):
    nf = config.model.normalize_factors
    if isinstance(nf, str):
        nf = [int(x) for x in nf.split(",")]
    nf = list(nf)
    return nf + [1] if len(nf) == 3 else nf


def get_data_inverse_scaler(config):
    """Closure equivalent to utils.py:71-105 (tensor plumbing for callers outside the fused path)."""
    pos_norm, atom_norm, fc_norm, edge_norm = _factors(config)
    centered = config.data.centered

    def inverse_scale_fn(pos, atom_type, fc_charge, node_mask, edge_type=None, edge_mask=None):
        if pos is not None:
            pos = pos * pos_norm * node_mask
        atom_type = atom_type * atom_norm
        fc_charge = fc_charge * fc_norm * node_mask
        if centered:
            atom_type = (atom_type + 1.0) / 2.0 * node_mask
        if edge_type is not None:
            edge_type = edge_type * edge_norm
            if centered:
                edge_type = (edge_type + 1.0) / 2.0
            edge_type = edge_type * edge_mask.reshape(node_mask.size(0), node_mask.size(1), node_mask.size(1), 1)
            return pos, atom_type, fc_charge, edge_type
        return pos, atom_type, fc_charge

    inverse_scale_fn.factors = (pos_norm, atom_norm, fc_norm, edge_norm)
    inverse_scale_fn.centered = centered
    return inverse_scale_fn


def get_self_cond_fn(config):
    """utils.py:108-150.  'ori' hands the prediction on unchanged; 'clamp' clamps the predicted atom-type and charge
    channels IN PLACE (the clamped prediction therefore also enters the posterior mean, sampling.py:590,604-606) and
    returns a clamped copy of the edge prediction.  Elementwise torch ops on the device tensors the HIP forward wrote."""
    process_type = config.model.self_cond_type
    atom_types, include_fc = config.data.atom_types, config.model.include_fc_charge
    _, atom_norm, fc_norm, edge_norm = _factors(config)
    lo, hi = (-1.0, 1.0) if config.data.centered else (0.0, 1.0)
    fc_lo, fc_hi = (float(v) / fc_norm for v in config.data.fc_scale)

    def process_self_cond(cond_x, cond_edge_x):
        if process_type == "ori":
            return cond_x, cond_edge_x
        if process_type == "clamp":
            cond_x[:, :, 3:3 + atom_types].clamp_(lo / atom_norm, hi / atom_norm)
            if include_fc:
                cond_x[:, :, -1:].clamp_(fc_lo, fc_hi)
            return cond_x, cond_edge_x.clamp(lo / edge_norm, hi / edge_norm)
        raise ValueError("Self-condition data process error.")

    process_self_cond.in_place_clamp = process_type != "ori"    # the sampler keeps such hooks out of graph replay
    return process_self_cond


def hip_post_process_supported(config) -> bool:
    """ds_post_process hard-codes normalize_factors (1,4,4,1), centered, compress_edge, 5 atom types + charge."""
    return (tuple(_factors(config)) == (1, 4, 4, 1) and config.data.centered and config.data.compress_edge
            and config.data.atom_types == 5 and config.model.include_fc_charge and config.model.edge_ch == 2)
