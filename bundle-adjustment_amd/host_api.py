"""Python face of the C++ host mirror (``host/_jaicov_host``): JAICOV's Camera / Image / ObjectCoordinate /
BundleAdjustment API with the reference's names, bound with pybind11.  Importing this module needs the built
extension (``make -C bundle-adjustment_amd/host``); the extension links the HIP library, there is no CPU fallback."""
import glob
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    cands = glob.glob(os.path.join(_HERE, "host", "_jaicov_host*.so"))
    if not cands:
        raise ImportError("host/_jaicov_host*.so is missing: run __graft_entry__.build()")
    spec = importlib.util.spec_from_file_location("_jaicov_host", cands[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_m = _load()
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("_")})


def flat_problem(adjustment):
    """FlatProblem from a prepared + flattened BundleAdjustment (parity tests hand it to the oracle)."""
    import numpy as np
    from .problem import FlatProblem
    d = adjustment.flat()
    nb = d["blk_ip_begin"]
    return FlatProblem(
        n_unknowns=int(d["n_unknowns"]), rank_defect=int(d["rank_defect"]), datum_flags=int(d["datum_flags"]),
        point_col=d["point_col"], point_datum=d["point_datum"], io_col=d["io_col"], cam_r0=d["cam_r0"],
        cam_dist_begin=d["cam_dist_begin"], dist_kind=d["dist_kind"], dist_order=d["dist_order"], dist_col=d["dist_col"],
        image_camera=d["image_camera"], eo_col=d["eo_col"], ip_image=d["ip_image"], ip_point=d["ip_point"],
        ip_x=d["ip_x"], ip_y=d["ip_y"], ip_var_x=d["ip_var_x"], ip_var_y=d["ip_var_y"], ip_rho=d["ip_rho"],
        values=d["values"], sigma2apriori=float(d["sigma2apriori"]), blk_ip_begin=nb if len(nb) else np.zeros(1, np.int32),
        blk_disp_offset=d["blk_disp_offset"], blk_disp=d["blk_disp"], sb_point_a=d["sb_point_a"], sb_point_b=d["sb_point_b"],
        sb_length=d["sb_length"], sb_var=d["sb_var"], dg_row_begin=d["dg_row_begin"], dg_slot=d["dg_slot"],
        dg_obs=d["dg_obs"], dg_var=d["dg_var"], dg_disp_offset=d["dg_disp_offset"], dg_disp=d["dg_disp"],
        n_observations=adjustment.getNumberOfObservations())
