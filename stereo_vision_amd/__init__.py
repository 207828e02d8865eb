"""sgmx -- MI355X-native semi-global matching behind the reference's SGM interface.

Only the dense-stereo hot path of ivankreso/stereo-vision lives here: the HIP kernels
and C ABI (csrc/, include/sgmx.h -> libsgmx.so) and the host-side mirror of
recon::StereoSGM / recon::SGM (sgm.py)."""
from .sgm import (SGM, SGMParams, SGMStereo, StereoSGM, StereoSGMParams, load_data_cost, load_representation,  # noqa: F401
                  save_representation)
from ._lib import FLAG_MATERIALIZE, SgmxError  # noqa: F401
