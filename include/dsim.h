/* dsim.h -- C ABI of the MI355X-native differentiable articulated rigid-body step.
 *
 * This is the drop-in boundary for the hot path of NVlabs/DiffRL's dflex:
 *
 *   dsim_step_forward   replaces  SimulateFunc.forward  (dflex/dflex/sim.py:2097-2123), i.e. the
 *                       `substeps` calls of SemiImplicitIntegrator._simulate (sim.py:2225-2601)
 *                       that one SemiImplicitIntegrator.forward (sim.py:2182-2221) performs:
 *                       eval_rigid_fk, eval_rigid_id, eval_rigid_contacts_art, eval_muscles,
 *                       eval_rigid_tau, eval_rigid_jacobian, eval_rigid_mass,
 *                       eval_dense_gemm_batched x2, eval_dense_cholesky_batched,
 *                       eval_dense_solve_batched, eval_rigid_integrate
 *   dsim_step_backward  replaces  SimulateFunc.backward (sim.py:2127-2154) + Tape.replay
 *                       (dflex/dflex/adjoint.py:2153-2199): the reverse sweep over the same launches
 *   dsim_model_create   replaces  the tensors ModelBuilder.finalize + Model.collide upload
 *                       (dflex/dflex/model.py:1646-1879, 424-515) -- but ONE articulation template
 *                       shared by all environments instead of N replicated copies
 *
 * Conventions (same as the reference FFI, dflex/dflex/adjoint.py:1250-1289, where noted):
 *   - plain pointers + sizes; device pointers are borrowed for the duration of the call;
 *   - all floating point is IEEE fp32, indices int32;
 *   - state tensors are environment-major: q[N][n_q], qd[N][n_qd], act[N][n_qd],
 *     muscle_act[N][n_muscles]  (== the reference's flat joint_q.view(N, -1));
 *   - spatial_transform = (px,py,pz, qx,qy,qz,qw); spatial_vector = (wx,wy,wz, vx,vy,vz);
 *     6x6 inertia row-major, rotational block upper-left (dflex/dflex/util.py:340-349);
 *   - kernels are enqueued on the caller's HIP stream, no host synchronisation inside;
 *   - gradients are WRITTEN (not accumulated) -- unlike the reference (adjoint.h:335-347) the
 *     caller does not need to zero them;
 *   - every function returns DSIM_OK or a negative error code; dsim_last_error() gives the text.
 *     (The reference has no error returns: with_pytorch_error_handling=False, adjoint.py:1875.)
 */
#ifndef DSIM_H
#define DSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSIM_OK 0
#define DSIM_ERR_INVALID (-1)   /* bad argument / unsupported model */
#define DSIM_ERR_HIP (-2)       /* HIP runtime error */
#define DSIM_ERR_LIMIT (-3)     /* model exceeds a compiled-in limit */

/* joint types, dflex/dflex/model.py:26-31 */
#define DSIM_JOINT_PRISMATIC 0
#define DSIM_JOINT_REVOLUTE 1
#define DSIM_JOINT_BALL 2
#define DSIM_JOINT_FIXED 3
#define DSIM_JOINT_FREE 4

/* One articulation ("environment") template. All pointers are HOST pointers and are copied. */
typedef struct dsim_model_desc {
    int32_t n_links;      /* L */
    int32_t n_q;          /* generalized coordinates per articulation */
    int32_t n_qd;         /* degrees of freedom per articulation */
    int32_t n_contacts;   /* static ground-contact points (Model.collide) ; 0 if ground is off */
    int32_t n_muscles;    /* M */
    int32_t n_waypoints;  /* W = muscle_start[M] */
    const int32_t* joint_type;      /* [L] */
    const int32_t* joint_parent;    /* [L]  -1 = world; parents precede children */
    const int32_t* joint_q_start;   /* [L+1] */
    const int32_t* joint_qd_start;  /* [L+1] */
    const float* joint_X_pj;        /* [L][7] joint frame in parent */
    const float* joint_X_cm;        /* [L][7] body COM frame in child (rotation is identity) */
    const float* joint_axis;        /* [L][3] */
    const float* body_I_m;          /* [L][36] spatial inertia at the COM */
    const float* joint_armature;    /* [n_qd] */
    const float* joint_target;      /* [n_q] */
    const float* joint_target_ke;   /* [L] */
    const float* joint_target_kd;   /* [L] */
    const float* joint_limit_lower; /* [n_q] */
    const float* joint_limit_upper; /* [n_q] */
    const float* joint_limit_ke;    /* [L] */
    const float* joint_limit_kd;    /* [L] */
    const int32_t* contact_body;    /* [C] */
    const float* contact_point;     /* [C][3] */
    const float* contact_dist;      /* [C] */
    const float* contact_material;  /* [C][4] (ke, kd, kf, mu) already gathered per contact */
    const int32_t* muscle_start;    /* [M+1] */
    const int32_t* muscle_links;    /* [W] */
    const float* muscle_points;     /* [W][3] */
    float gravity[3];
} dsim_model_desc;

typedef struct dsim_model dsim_model; /* opaque, owns device copies of the template */

const char* dsim_last_error(void);
int dsim_version(void);

/* Uploads the template to the CURRENT HIP device; the model belongs to that device from then on.  One process may hold
 * models on several GPUs ("one stream (or process) per GPU"): every call that takes a model must be
 * made with the model's device current (hipSetDevice / torch.cuda.device), with a stream and pointers of that device, and
 * returns DSIM_ERR_INVALID otherwise -- a launch on another device would read a constant block that is not there. */
int dsim_model_create(const dsim_model_desc* desc, dsim_model** out);
int dsim_model_destroy(dsim_model* m);
int dsim_model_device(const dsim_model* m);   /* HIP device index the model was created on */
/* 0: generic kernels (runtime layout); > 0: a per-model specialised kernel set is in use (layout matched a
 * compile-time table exactly).  Same results either way. */
int dsim_model_variant(const dsim_model* m);

/* Floats per environment of the checkpoint the forward launch leaves in HBM for the adjoint launch: per
 * substep one "saved block" (q, qd and the forward intermediates the adjoint reads: transforms, motion
 * subspace, twists, inertias, subtree forces, accelerations) plus one inverse mass matrix per refresh.
 * (The reference keeps all 14 State tensors of every substep alive on its Tape, sim.py:2111 /
 * model.py:338-392.)  dsim_ckpt_floats() is the upper bound over mm_freq (mm_freq = 1). */
int64_t dsim_ckpt_floats_mm(const dsim_model* m, int substeps, int mm_freq);

/* Checkpoint mode of a model (default DSIM_CKPT_FULL).  Affects dsim_ckpt_floats(_mm) and every later forward / backward
 * call with a checkpoint; a checkpoint must be consumed in the mode it was written in.
 *   DSIM_CKPT_FULL  one row per substep with everything the adjoint reads (q, qd, X_sc, S, v, a, world inertias, f_tot,
 *                   qdd): the adjoint launch reads it back instead of recomputing it (fastest; HBM is idle on this path).
 *                   Floats per environment and env-step: Ant 7,524 (30 KB), Humanoid 49,748 (199 KB), SNUHumanoid
 *                   33,272 (133 KB), i.e. 0.99 / 6.5 / 2.2 GB for a 1024- (SNU: 512-) environment, H = 32 rollout.
 *   DSIM_CKPT_LEAN  one row per substep with (q, qd) only + the inverse mass matrices: the adjoint launch recomputes the
 *                   forward phases of every substep (same code, bit-identical intermediates, identical gradients).
 *                   Ant 740 floats (3 KB), Humanoid 3,476 (14 KB), SNUHumanoid 6,200 (25 KB) per environment and env-step:
 *                   for rollouts whose full checkpoint would not fit (tens of thousands of environments per GPU). */
#define DSIM_CKPT_FULL 0
#define DSIM_CKPT_LEAN 1
int dsim_model_set_ckpt_mode(dsim_model* m, int mode);
int64_t dsim_ckpt_floats(const dsim_model* m, int substeps);

/* One env.step() worth of simulation for N environments: `substeps` semi-implicit substeps of
 * dt/substeps with joint_act / muscle_act held fixed; mass matrix + Cholesky factor refreshed on
 * substeps i with i % mm_freq == 0 (sim.py:2113).
 *   ckpt: [N][dsim_ckpt_floats_mm(m, substeps, mm_freq)] or NULL (no-grad fast path == dflex.config.no_grad,
 *   sim.py:2201); every row starts with the (q, qd) entering its substep
 *   muscle_act may be NULL when n_muscles == 0.  q_out/qd_out may alias q_in/qd_in.
 *
 * PRECONDITION -- THE PATH IS DEFINED ON UNIT QUATERNIONS ONLY, FORWARD AND ADJOINT.  Every quaternion block of q_in (free
 * joint: q[cs+3 .. cs+6], ball joint: q[cs .. cs+3]) must be a unit quaternion to 1e-4.  The reference evaluates its rotation
 * formulas literally off the manifold (quat.h:113-116 rotate, spatial.h:559-586 / sim.py:1117-1134 T^T I_m T with a
 * non-orthogonal R); this library uses forms that agree with them ON the manifold only (10-parameter rigid-body inertia, pose
 * cotangents as wrenches).  Measured on the Ant step recording with the root quaternion scaled (host harness vs the
 * reference-order oracle): |q| = 1.001 -> 5.3e-3 relative error in qd_out, 1.01 -> 4.7e-2, 1.1 -> 0.64, against the 1e-4
 * state tolerance on the manifold.  States produced by the path itself always qualify: the integrator renormalises every
 * quaternion in every substep (sim.py:1552, 1616), so only states a caller constructs (reset_with_state, hand-made inputs)
 * can violate this.
 * ENFORCEMENT: the forward kernels check the state as loaded, once per launch; | |q|^2 - 1 | > 2e-4 in any block of any
 * environment marks the model, and the NEXT dsim_* call that takes the model returns DSIM_ERR_INVALID (once; the mark is
 * cleared) instead of launching -- asynchronous, like the error of a HIP kernel, no synchronisation on the step path.
 * dsim_model_status() asks explicitly (synchronise the stream first for a definitive answer about launches in flight). */
int dsim_step_forward(const dsim_model* m, int n_envs,
                      const float* q_in, const float* qd_in, const float* act, const float* muscle_act,
                      float dt, int substeps, int mm_freq,
                      float* q_out, float* qd_out, float* ckpt, void* hip_stream);

/* DSIM_OK, or DSIM_ERR_INVALID if a forward launch since the last report was handed a non-unit quaternion (see the
 * precondition above); *first_env (may be NULL) receives an environment index that saw one, -1 otherwise.  Reads two words
 * of host memory the kernels write to; never touches a stream.  Reporting clears the mark. */
int dsim_model_status(dsim_model* m, int* first_env);

/* Reverse sweep of the same step.  Needs the checkpoint of the forward call and the same act /
 * muscle_act.  Outputs: gq_in[N][n_q], gqd_in[N][n_qd], gact[N][n_qd], gmuscle_act[N][M]
 * (gact / gmuscle_act may be NULL to skip).  Follows the reference's adjoint conventions
 * (SURVEY.md App. B): Cholesky treated as constant, dH = -(LL^T)^-1 g_qdd * qdd^T accumulated over
 * the substeps that reuse a factor (matnn.h:310-336), min/max/clamp/step/normalize rules of
 * adjoint.h:129-190 and vec3.h:204-222.
 *
 * Defined on unit quaternions only, like the forward call whose checkpoint it consumes (precondition at dsim_step_forward).
 * QUATERNION COORDINATES, TWO FORMS.  For every quaternion block of joint_q (free joint: q[cs+3 .. cs+6], ball joint:
 * q[cs .. cs+3]) the reference's SimulateFunc.backward (sim.py:2127-2154) returns a cotangent WITH a component along the
 * quaternion itself: it differentiates its rotation formulas literally (quat.h:232-288 adj_mul / adj_rotate,
 * spatial.h:740-798), also in the direction in which |quat| changes, where those formulas are not rotations.  That "radial"
 * component is annihilated by the integrator's quaternion normalisation (sim.py:1552, 1616) in every upstream propagation --
 * gradients of rollouts w.r.t. actions do not depend on it -- and the adjoint kernels, where a pose cotangent is a world-frame
 * wrench, do not produce it:
 *   dsim_step_backward           gq_in == reference's gq_in minus, per quaternion block, (u . g_ref) u with u = quat / |quat|
 *                                (32 % / 16 % / 8 % of max |gq_in| on the Ant / Humanoid / SNUHumanoid recordings); every other
 *                                output (all non-quaternion coordinates of gq_in, gqd_in, gact, gmuscle_act) equals the
 *                                reference's to the fp32 tolerance.  The hot path: what DFlexEnv.step and SHAC use.
 *   dsim_step_backward_literal   gq_in == reference's gq_in, quaternion blocks included (measured against the reference's
 *                                recordings, UN-projected: 1.1e-6 / 5.8e-6 / 7.4e-6 max-norm relative, Ant / Humanoid /
 *                                SNUHumanoid).  The same adjoint launch + one small launch that evaluates the radial component
 *                                rho_j = dL/d eps_j under q_j -> (1 + eps_j) q_j in forward mode through the reference's literal
 *                                formulas of the first substep (csrc/dsim_literal.hpp) and adds rho_j q_j.  scratch:
 *                                [N][dsim_literal_scratch_floats(m)] floats of device memory the two launches hand over in
 *                                (the first substep's output cotangents and the first mass-matrix group's cotangent).
 *                                Models of up to 24 links / 28 dofs (DSIM_ERR_LIMIT beyond).
 * tests/test_gpu_parity.py asserts both statements on the HIP kernels' raw output. */
int dsim_step_backward(const dsim_model* m, int n_envs,
                       const float* ckpt, const float* act, const float* muscle_act,
                       float dt, int substeps, int mm_freq,
                       const float* gq_out, const float* gqd_out,
                       float* gq_in, float* gqd_in, float* gact, float* gmuscle_act, void* hip_stream);
int64_t dsim_literal_scratch_floats(const dsim_model* m);
int dsim_step_backward_literal(const dsim_model* m, int n_envs,
                               const float* ckpt, const float* act, const float* muscle_act,
                               float dt, int substeps, int mm_freq,
                               const float* gq_out, const float* gqd_out,
                               float* gq_in, float* gqd_in, float* gact, float* gmuscle_act, float* scratch, void* hip_stream);

/* Derived body transforms of a joint state: X_sc[N][L][7] (link frames in the world, what eval_rigid_fk writes to
 * State.body_X_sc, sim.py:1638-1678) and, if X_sm is not NULL, X_sm[N][L][7] = X_sc o X_cm (State.body_X_sm: the bodies'
 * centre-of-mass frames).  The reference's State carries them after every forward() (model.py:338-392; read by
 * envs/snu_humanoid.py:209,227 for rendering) -- computed from the joint coordinates ENTERING the last substep; the
 * step functions above keep them in LDS / the checkpoint only, so this is the read-back: pass the q of interest (for the
 * reference's literal value: the head of the last substep's checkpoint row, which is that substep's input q). */
int dsim_body_transforms(const dsim_model* m, int n_envs, const float* q, float* X_sc, float* X_sm, void* hip_stream);

/* ---- fused environment surface -------------------------------------
 * The per-step torch glue of the reference environments -- action clip + scale into joint_act /
 * muscle activations (envs/ant.py:157-163, humanoid.py:188-211, snu_humanoid.py:245-271,
 * cartpole_swing_up.py:115-120), calculateObservations and calculateReward (ant.py:266-303,
 * humanoid.py:314-354, snu_humanoid.py:376-414, cartpole_swing_up.py:204-222) -- evaluated inside the
 * step kernels, adjoint included, so that one env.step() is ONE launch forward and ONE backward. */
#define DSIM_ENV_LOCOMOTION 1 /* free root: [h, quat(4), lin vel(3), ang vel(3), q[7:], s*qd[6:], up, heading, (actions)] */
#define DSIM_ENV_CARTPOLE 2   /* [x, xdot, sin th, cos th, thdot] */
#define DSIM_ENV_PLANAR 3     /* planar root (hopper, half-cheetah; envs/hopper.py:273, cheetah.py:254): [q[1:], qd] */
#define DSIM_REW_ANT 0
#define DSIM_REW_HUMANOID 1
#define DSIM_REW_SNU 2
#define DSIM_REW_CARTPOLE 3
#define DSIM_REW_HOPPER 4     /* cartpole_penalties[0] carries the termination angle */
#define DSIM_REW_CHEETAH 5

typedef struct dsim_env_spec {
    int32_t kind, rew_kind;
    int32_t n_act, n_obs;
    int32_t act_offset;   /* joint_act[act_offset + k] = clip(a_k,-1,1) * act_scale[k]        (act_muscle == 0) */
    int32_t act_muscle;   /* muscle_act[k] = (clip(a_k,-1,1) * 0.5 + 0.5) * act_scale[k]      (act_muscle == 1) */
    int32_t obs_actions;  /* observation ends with the stored actions */
    int32_t sanitize_grads; /* dsim_env_step_backward writes 0 for every non-finite cotangent it would return (gq_in, gqd_in,
                             * gactions): the nan_to_num(grad, 0, 0, 0) hooks that the reference's humanoid environments register on
                             * state.joint_q / joint_qd / actions in every step (envs/humanoid.py:195-206, snu_humanoid.py:253-264),
                             * done in the adjoint launch's own output stores instead of three extra torch kernels per step (ABI 106) */
    float inv_start_rot[4];
    float target_x, target_z;      /* targets + start_pos */
    float termination_height, termination_tolerance, height_rew_scale, action_penalty, joint_vel_obs_scaling;
    float cartpole_penalties[4];   /* pole angle, pole velocity, cart position, cart velocity */
    const float* act_scale;        /* [n_act] DEVICE pointer, borrowed per call */
} dsim_env_spec;

/* Episode bookkeeping fused into the step kernel.  Replaces, per env.step(), the torch ops and the device->host sync of
 * the reference: progress_buf += 1 and reset_buf at the end of calculateReward (envs/ant.py:176-184, 297-307;
 * humanoid.py:340-356; hopper.py:288-293), env_ids = reset_buf.nonzero() and reset(env_ids) (ant.py:186-234).
 * A finished environment e restarts from entry (reset_count[e] % reset_pool) of a pool of start states [normally one:
 * the deterministic start state], perturbed IN THE KERNEL when noise_q / noise_qd are given -- a fresh draw for every
 * restart, as the reference's reset() does with torch.rand (envs/ant.py:199-234, hopper.py:180-212,
 * cartpole_swing_up.py:140-160): coordinate k gets  + noise[k] * (u - 0.5)  with u uniform in [0, 1) from the counter-based
 * generator Philox4x32-10 keyed by `seed` at counter (environment index, reset_count[e], coordinate), so a restart never
 * repeats a state, needs no host work and is replay-safe under a captured HIP graph; for a free-floating root
 * (DSIM_ENV_LOCOMOTION) the start rotation is composed with a rotation by (u - 0.5) * noise_angle about a uniformly drawn
 * axis (normalize(u3 - 0.5), ant.py:208-213) and noise_q[3..6] is ignored.
 * q_out / qd_out / obs then describe the NEW state (stored actions cleared), obs_before_reset the old one, rew the old
 * one (0 for an invalid state), progress[e] = 0, done[e] = 1. */
typedef struct dsim_episode {
    int64_t* progress;          /* [N] in/out: progress_buf */
    int64_t* done;              /* [N] out: reset_buf */
    float* obs_before_reset;    /* [N][n_obs] out, may be NULL */
    const float* reset_q;       /* [reset_pool][N][n_q] */
    const float* reset_qd;      /* [reset_pool][N][n_qd] */
    int32_t* reset_count;       /* [N] in/out: restarts so far */
    int32_t reset_pool;
    int32_t episode_length;     /* done when progress > episode_length - 1 */
    int32_t height_terminate;   /* done when obs[0] < spec.termination_height */
    int32_t check_invalid;      /* done, reward 0, when obs / q / qd is non-finite or |q|, |qd| > 1e6 */
    const float* noise_q;       /* [n_q] DEVICE pointer or NULL: amplitude of the uniform restart noise per coordinate */
    const float* noise_qd;      /* [n_qd] DEVICE pointer or NULL */
    float noise_angle;          /* free-floating root: amplitude of the random start rotation (radians) */
    uint64_t seed;              /* Philox key */
} dsim_episode;

/* actions[N][n_act] -> q_out, qd_out, obs[N][n_obs], rew[N]  (+ ckpt as in dsim_step_forward; use dsim_ckpt_floats_mm).
 * episode may be NULL: plain step, no bookkeeping.  Same precondition and enforcement as dsim_step_forward: q_in holds
 * unit quaternions (the states this function returns, restarts included, always do). */
int dsim_env_step_forward(const dsim_model* m, const dsim_env_spec* env, int n_envs,
                          const float* q_in, const float* qd_in, const float* actions,
                          float dt, int substeps, int mm_freq,
                          float* q_out, float* qd_out, float* obs, float* rew, float* ckpt,
                          const dsim_episode* episode, void* hip_stream);

/* cotangents (gq_out, gqd_out, gobs, grew, gobs_before_reset) -> (gq_in, gqd_in, gactions).  Any of the five inputs may
 * be NULL (= zeros).  For an environment the forward step restarted, gq_out / gqd_out / gobs refer to the new state and
 * are ignored, as autograd does in the reference after reset()'s in-place writes. */
int dsim_env_step_backward(const dsim_model* m, const dsim_env_spec* env, int n_envs,
                           const float* ckpt, const float* actions,
                           float dt, int substeps, int mm_freq,
                           const float* gq_out, const float* gqd_out, const float* gobs, const float* grew,
                           const float* gobs_before_reset,
                           float* gq_in, float* gqd_in, float* gactions, void* hip_stream);

/* observation + reward of a given state with given stored actions (reset / initialize_trajectory path) */
int dsim_env_observe(const dsim_model* m, const dsim_env_spec* env, int n_envs, const float* q, const float* qd,
                     const float* stored_actions, float* obs, float* rew, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* DSIM_H */
