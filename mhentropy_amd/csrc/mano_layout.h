// Layout (in floats) of the packed MANO table blob shared by the host packer
// (mhentropy_amd/mano_pack.py mirrors these numbers; a test compares them) and
// the kernels in mano.hip.  The first JOINT_FLOATS floats are everything the
// loss pass needs and are staged into LDS; the vertex section is read
// lane-per-vertex straight from L2.
#pragma once
namespace mhe { namespace mano {
constexpr int NV = 778;
constexpr int VP = 832;           // vertices padded to 13 wavefronts
// ---- joint section -------------------------------------------------------
constexpr int COMPS = 0;          // [45][45]  th_selected_comps[k][l]      (manolayer.py:137)
constexpr int MEAN = 2028;        // [45]      th_hands_mean                (manolayer.py:143)
constexpr int JT = 2076;          // [16][3]   J_regressor @ v_template     (manolayer.py:181-184)
constexpr int JSD = 2124;         // [16][3][10] J_regressor @ shapedirs
constexpr int TIP_T = 2604;       // [5][3]    v_template of the tip vertices (manolayer.py:251)
constexpr int TIP_SD = 2620;      // [5][3][10]
constexpr int TIP_PD = 2772;      // [5][3][135]
constexpr int TIP_W = 4800;       // [5][16]   skinning weights of the tips
constexpr int JOINT_FLOATS = 4880;
// ---- vertex section (vertex index fastest) --------------------------------
constexpr int V_T = JOINT_FLOATS;          // [3][VP]
constexpr int V_SD = V_T + 3 * VP;         // [10][3][VP]
constexpr int V_PD = V_SD + 30 * VP;       // [135][3][VP]
constexpr int V_W = V_PD + 405 * VP;       // [16][VP]
constexpr int V_JR = V_W + 16 * VP;         // [16][VP]   J_regressor (wrapper re-regression, ManoLayer.py:141-148)
constexpr int TOTAL_FLOATS = V_JR + 16 * VP;
// ---- per-hypothesis workspace row of the full-mesh pass (mano_pose_kernel -> the skinning kernels) --------
// pose map (135), beta (10), the 16 skinning transforms [j][12] (192), centre / root / bone (7)
constexpr int WS_PM = 0, WS_BT = 135, WS_GR = 145, WS_NRM = 337, WS_STRIDE = 352;
}}  // namespace
