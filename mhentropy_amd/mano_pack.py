"""Host-side packing of the MANO model buffers into the table blob the kernels
read (layout constants mirror csrc/mano_layout.h; tests compare them with
mhe_mano_table_floats()).  Pure data movement plus two constant folds that do not
depend on any input:  J_regressor @ v_template and J_regressor @ shapedirs
(the rest joints are affine in beta, reference hand/manopth/manolayer.py:181-184).
"""
import numpy as np

NV, VP = 778, 832
COMPS, MEAN, JT, JSD, TIP_T, TIP_SD, TIP_PD, TIP_W, JOINT_FLOATS = 0, 2028, 2076, 2124, 2604, 2620, 2772, 4800, 4880
V_T = JOINT_FLOATS
V_SD = V_T + 3 * VP
V_PD = V_SD + 30 * VP
V_W = V_PD + 405 * VP
V_JR = V_W + 16 * VP
TOTAL_FLOATS = V_JR + 16 * VP

TIP_VERTS_RIGHT = (745, 317, 444, 556, 673)       # reference hand/manopth/manolayer.py:251


def pack_tables(shapedirs, posedirs, v_template, J_regressor, weights, selected_comps, hands_mean):
    """All arguments numpy; shapes as the reference buffers
    (manolayer.py:69-101): shapedirs (778,3,10), posedirs (778,3,135),
    v_template (778,3), J_regressor (16,778), weights (778,16),
    selected_comps (45,45), hands_mean (45,)."""
    f64 = lambda a: np.asarray(a, np.float64)
    sd, pd, vt, jr, w = f64(shapedirs), f64(posedirs), f64(v_template).reshape(NV, 3), f64(J_regressor), f64(weights)
    blob = np.zeros(TOTAL_FLOATS, np.float32)
    blob[COMPS:COMPS + 2025] = np.asarray(selected_comps, np.float32).reshape(-1)
    blob[MEAN:MEAN + 45] = np.asarray(hands_mean, np.float32).reshape(-1)
    blob[JT:JT + 48] = (jr @ vt).astype(np.float32).reshape(-1)
    blob[JSD:JSD + 480] = np.einsum("jv,vck->jck", jr, sd).astype(np.float32).reshape(-1)
    tips = list(TIP_VERTS_RIGHT)
    blob[TIP_T:TIP_T + 15] = vt[tips].astype(np.float32).reshape(-1)
    blob[TIP_SD:TIP_SD + 150] = sd[tips].astype(np.float32).reshape(-1)
    blob[TIP_PD:TIP_PD + 2025] = pd[tips].astype(np.float32).reshape(-1)
    blob[TIP_W:TIP_W + 80] = w[tips].astype(np.float32).reshape(-1)

    def vsec(a):      # (778, X) -> (X, VP) vertex-fastest, zero padded
        out = np.zeros((a.shape[1], VP), np.float32)
        out[:, :NV] = a.T
        return out.reshape(-1)
    blob[V_T:V_SD] = vsec(vt)
    blob[V_SD:V_PD] = vsec(sd.transpose(0, 2, 1).reshape(NV, 30))       # [k][c][v]
    blob[V_PD:V_W] = vsec(pd.transpose(0, 2, 1).reshape(NV, 405))
    blob[V_W:V_JR] = vsec(w)
    blob[V_JR:TOTAL_FLOATS] = vsec(jr.T)
    return blob
