/*
 * oracle/twoarmy_oracle.c -- CPU restatement of the reference's MiniGrid-Twoarmy hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP engine in
 * <package>/csrc and the `cpu_baseline` ("port") leg of bench.py.  It is never linked into,
 * imported by, or used as a fallback for the product path.
 *
 * Parity pin: checked bit-for-bit against golden vectors recorded from the reference itself
 * (imported in the build container by oracle/gen_golden.py; fixtures in tests/golden/).
 *
 * The code follows the reference algorithm literally (object grid, slice + rotate_left loops)
 * instead of the closed forms the HIP engine uses, so that the two are independent derivations.
 * Reference citations are relative to /root/reference:
 *   G  = gym_minigrid/minigrid.py
 *   V6 = gym_minigrid/envs/twoarmy_v6.py
 *   V4 = gym_minigrid/envs/twoarmy_v4.py
 *   EB = soa/env_buffer.py
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GS 17               /* grid size (gym_minigrid/__init__.py:13,19  kwargs size=17) */
#define NC (GS * GS)

/* OBJECT_TO_IDX / COLOR_TO_IDX  (G:40-67) */
enum { T_UNSEEN = 0, T_EMPTY = 1, T_WALL = 2, T_FLOOR = 3, T_DOOR = 4, T_KEY = 5, T_BALL = 6,
       T_BOX = 7, T_GOAL = 8, T_LAVA = 9, T_AGENT = 10, T_SUBGOAL = 11 };
enum { C_RED = 0, C_GREEN = 1, C_BLUE = 2, C_PURPLE = 3, C_YELLOW = 4, C_GREY = 5 };

/* reward codes -> values (V6:181,232,243,287,298).  The reference compares `reward == -0.1`
 * (V6:290) on an assigned literal, so an integer code is exact. */
enum { R_STEP = 0, R_RISK = 1, R_HIT = 2, R_ROOM2 = 3, R_GOAL = 4 };
static const double REWARD_VALUE[5] = { -0.01, -0.1, -0.9, 0.2, 0.9 };

/* error codes: what the reference would have raised */
enum { E_OK = 0, E_ATTRIBUTE = 1 /* actions.forward missing, G:1397 */,
       E_ASSERT = 2 /* Grid.get/set bounds assert, G:599-607 */,
       E_TYPE = 3 /* cur_pos is None subscripted, V4:122-124 */ };

/* draw slots (one 32-bit word each; value = lo + word % n) */
enum { S_GATE = 0, S_WALL1 = 1, S_WALL2 = 2, S_SPAWN = 3, S_COIN_A = 4, S_COIN_B = 5, S_ACTION = 8 };
#define DRAW_TAG 0x54574F41u /* "TWOA" */

typedef struct {
    uint8_t type[NC], colour[NC], state[NC]; /* Grid.grid, index j*width+i (G:599-607); None == T_EMPTY */
    int32_t ax, ay, dir;                     /* agent_pos, agent_dir */
    int32_t step_count, max_steps;
    int32_t variant;                         /* 4 or 6 */
    int32_t step_move, pone, patrol, up1, right2, upd_long, upd_horiz, risk_count, first_to_room2;
    int32_t ob_x[3], ob_y[3];                /* obstacles[k].cur_pos */
    int32_t o1_x[3], o1_y[3], o1_valid;      /* obstacles1[k].cur_pos (None until spawned) */
    int32_t o2_x[4], o2_y[4], o2_valid;      /* obstacles2[k].cur_pos */
    int32_t goal_x, goal_y;
    uint32_t t;                              /* step() calls since construction: draw counter */
    int32_t error;
    /* outputs of the last step */
    int32_t reward_code, terminated, truncated;
} tw_env;

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

/* word for (seed, env_id, t, slot): counter = (env_id, t, slot>>2, TAG), word = slot&3 */
uint32_t tw_oracle_draw_word(uint64_t seed, uint32_t env_id, uint32_t t, uint32_t slot) {
    uint32_t c[4] = { env_id, t, slot >> 2, DRAW_TAG };
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
    return c[slot & 3];
}

/* ------------------------------------------------------------------ grid helpers */
static int in_bounds(int i, int j) { return i >= 0 && i < GS && j >= 0 && j < GS; }

/* Grid.set (G:599-602); returns 0 if the reference's assert would fire */
static int grid_set(tw_env *e, int i, int j, int type, int colour) {
    if (!in_bounds(i, j)) return 0;
    int k = j * GS + i;
    e->type[k] = (uint8_t)type; e->colour[k] = (uint8_t)colour; e->state[k] = 0;
    return 1;
}

/* can_overlap (G:291-293 default False; Goal/SubGoal/Floor/Lava True G:360,370,384,398; open Door) */
static int can_overlap(int type, int state) {
    return type == T_GOAL || type == T_SUBGOAL || type == T_FLOOR || type == T_LAVA ||
           (type == T_DOOR && state == 0);
}

/* Twoarmy_v6._gen_grid (V6:39-81) == Twoarmy_v4._gen_grid (V4:38-80) + MiniGridEnv.reset (G:947-980).
 * Leaves every Twoarmy flag alone (they are only re-armed by the episode-end block of step()). */
void tw_oracle_reset(tw_env *e) {
    for (int k = 0; k < NC; ++k) { e->type[k] = T_EMPTY; e->colour[k] = 0; e->state[k] = 0; }
    /* wall_rect(0,0,w,h)  G:621-625 */
    for (int i = 0; i < GS; ++i) {
        grid_set(e, i, 0, T_WALL, C_GREY); grid_set(e, i, GS - 1, T_WALL, C_GREY);
        grid_set(e, 0, i, T_WALL, C_GREY); grid_set(e, GS - 1, i, T_WALL, C_GREY);
    }
    for (int i = 1; i < 6; ++i) grid_set(e, i, 8, T_WALL, C_GREY);   /* V6:46-47 */
    for (int i = 11; i < 16; ++i) grid_set(e, i, 8, T_WALL, C_GREY); /* V6:48-49 */
    for (int k = 0; k < 3; ++k) {                                    /* V6:56-59 */
        grid_set(e, k + 7, 8, T_BALL, C_YELLOW);
        e->ob_x[k] = k + 7; e->ob_y[k] = 8;
    }
    e->o1_valid = 0; e->o2_valid = 0;        /* fresh Ball objects, cur_pos None (V6:58,62) */
    e->ax = 3; e->ay = 15; e->dir = 3;       /* V6:65-69 */
    grid_set(e, e->ax, e->ay, T_EMPTY, 0);
    e->goal_x = 14; e->goal_y = 2;           /* V6:73-77 */
    grid_set(e, e->goal_x, e->goal_y, T_GOAL, C_GREEN);
    e->step_count = 0;                       /* G:972 */
    e->error = E_OK;
}

/* Twoarmy_v{4,6}.__init__ (V6:10-37) followed by the reset() MiniGridEnv.__init__ ends with (G:945) */
void tw_oracle_init(tw_env *e, int variant) {
    memset(e, 0, sizeof(*e));
    e->variant = variant;
    e->max_steps = 50;
    e->step_move = 0; e->pone = 0; e->upd_horiz = 0; e->upd_long = 1; e->patrol = 0;
    e->up1 = 0; e->right2 = 1; e->risk_count = 0; e->first_to_room2 = 1;
    e->t = 0;
    tw_oracle_reset(e);
}

/* ------------------------------------------------------------------ observation
 * gen_obs_grid (G:1443-1478) + Grid.encode (G:749-772).  out[(i*V + j)*3 + c], i = x index.
 * Literal: slice (G:641-660), rotate_left dir+1 times (G:627-639), agent cell := None. */
typedef struct { uint8_t t, c, s; } ocell;

void tw_oracle_gen_obs(const tw_env *e, int V, uint8_t *out) {
    ocell a[GS * GS], b[GS * GS]; /* V <= 17 */
    int topX, topY, h = V / 2;
    switch (e->dir) {            /* get_view_exts G:1262-1293 */
    case 0: topX = e->ax;          topY = e->ay - h;      break;
    case 1: topX = e->ax - h;      topY = e->ay;          break;
    case 2: topX = e->ax - V + 1;  topY = e->ay - h;      break;
    default: topX = e->ax - h;     topY = e->ay - V + 1;  break;
    }
    for (int j = 0; j < V; ++j)
        for (int i = 0; i < V; ++i) {
            int x = topX + i, y = topY + j;
            ocell v;
            if (in_bounds(x, y)) { int k = y * GS + x; v.t = e->type[k]; v.c = e->colour[k]; v.s = e->state[k]; }
            else { v.t = T_WALL; v.c = C_GREY; v.s = 0; }
            a[j * V + i] = v;
        }
    ocell *src = a, *dst = b;
    for (int r = 0; r < e->dir + 1; ++r) {   /* grid.set(j, height-1-i, get(i,j)) */
        for (int i = 0; i < V; ++i)
            for (int j = 0; j < V; ++j)
                dst[(V - 1 - i) * V + j] = src[j * V + i];
        ocell *tmp = src; src = dst; dst = tmp;
    }
    { ocell none = { T_EMPTY, 0, 0 }; src[(V - 1) * V + V / 2] = none; } /* G:1472-1476, carrying None */
    for (int i = 0; i < V; ++i)
        for (int j = 0; j < V; ++j) {
            ocell v = src[j * V + i];
            uint8_t *o = out + (i * V + j) * 3;
            o[0] = v.t; o[1] = v.c; o[2] = v.s;
        }
}

/* Env_transact.matrix_env (EB:300-318): m[j*17+i]; agent cell 0.3 */
void tw_oracle_matrix(const tw_env *e, float *m) {
    for (int k = 0; k < NC; ++k) {
        float v = 0.9f;
        if (e->type[k] == T_WALL) v = -0.9f;
        else if (e->type[k] == T_BALL) v = -0.5f;
        m[k] = v;
    }
    m[GS * e->ay + e->ax] = 0.3f;
}

/* Env_transact.data_env (EB:320-334): agent (y,x) and goal (y,x) as floats */
void tw_oracle_pos(const tw_env *e, float *agent_yx, float *goal_yx) {
    agent_yx[0] = (float)e->ay; agent_yx[1] = (float)e->ax;
    goal_yx[0] = (float)e->goal_y; goal_yx[1] = (float)e->goal_x;
}

/* ------------------------------------------------------------------ step */
static uint32_t take_draw(const uint32_t *draws, uint64_t seed, uint32_t env_id, uint32_t t, int slot) {
    return draws ? draws[slot] : tw_oracle_draw_word(seed, env_id, t, (uint32_t)slot);
}

/* move one patrol group: clear all cells, then put each at +d inside try/except (V4:119-176) */
static int move_group(tw_env *e, int n, int32_t *xs, int32_t *ys, int valid, int dx, int dy) {
    if (!valid) { e->error = E_TYPE; return 0; }   /* old_pos[0] on None */
    for (int k = 0; k < n; ++k)
        if (!grid_set(e, xs[k], ys[k], T_EMPTY, 0)) { e->error = E_ASSERT; return 0; }
    for (int k = 0; k < n; ++k) {
        int nx = xs[k] + dx, ny = ys[k] + dy;
        if (grid_set(e, nx, ny, T_BALL, C_YELLOW)) { xs[k] = nx; ys[k] = ny; } /* except: pass */
    }
    return 1;
}

/*
 * One Twoarmy_v{4,6}.step (V6:83-325 / V4:82-322) including MiniGridEnv.step (G:1333-1441).
 * `draws` (nullable): 8 words indexed by slot; NULL -> Philox(seed, env_id, e->t, slot).
 * `obs` (nullable): uint8[V*V*3] image produced by the gen_obs() inside MiniGridEnv.step,
 * i.e. BEFORE the wall drop / patrol spawn of the same step.
 * Returns the error code (0 = ok).  On error the state keeps whatever mutations the
 * reference had already made before raising.
 */
int tw_oracle_step(tw_env *e, int action, const uint32_t *draws, uint64_t seed, uint32_t env_id,
                   int V, uint8_t *obs) {
    const uint32_t t = e->t;
    e->t += 1;
    e->error = E_OK;
    if (action >= 7) action = 0;                       /* V6:85-86 (action_space.n == 7) */
    e->step_move += 1;                                 /* V6:88 */
    const int sm = e->step_move;

    /* row-8 balls (V6:96-112) */
    int old_x[3];
    for (int k = 0; k < 3; ++k) {
        if (!grid_set(e, e->ob_x[k], e->ob_y[k], T_EMPTY, 0)) { e->error = E_ASSERT; return e->error; }
        old_x[k] = e->ob_x[k];
    }
    for (int k = 0; k < 3; ++k) {
        int m6 = sm % 6, nx;
        if (m6 == 1 || m6 == 0) nx = old_x[k] + 1;
        else if (m6 == 2 || m6 == 3) nx = old_x[k] - 1;
        else nx = old_x[k];
        if (grid_set(e, nx, 8, T_BALL, C_YELLOW)) { e->ob_x[k] = nx; e->ob_y[k] = 8; } /* except: pass */
    }

    if (e->variant == 4) {
        /* V4:115-144 longitudinal patrol (obstacles1) */
        if (e->upd_long) {
            e->upd_horiz = 0;
            int go = (sm % 4 == 2) || (sm % 6 == 3) || (sm % 6 == 0);
            if (!go) go = (take_draw(draws, seed, env_id, t, S_GATE) % 10u) == 6u;
            if (go && e->patrol) {
                if (e->up1) {
                    if (!move_group(e, 3, e->o1_x, e->o1_y, e->o1_valid, 0, -1)) return e->error;
                    if (e->o1_y[0] == 3) e->up1 = 0;
                } else {
                    if (!move_group(e, 3, e->o1_x, e->o1_y, e->o1_valid, 0, +1)) return e->error;
                    if (e->o1_y[2] == 7) e->up1 = 1;
                }
            }
        }
        /* V4:147-176 horizontal patrol (obstacles2) */
        if (e->upd_horiz) {
            e->upd_long = 0;
            int go = (sm % 6 != 1);
            if (!go) go = (take_draw(draws, seed, env_id, t, S_GATE) % 10u) == 6u;
            if (go && e->patrol) {
                if (e->right2) {
                    if (!move_group(e, 4, e->o2_x, e->o2_y, e->o2_valid, +1, 0)) return e->error;
                    if (e->o2_x[3] == 11) e->right2 = 0;
                } else {
                    if (!move_group(e, 4, e->o2_x, e->o2_y, e->o2_valid, -1, 0)) return e->error;
                    if (e->o2_x[0] == 5) e->right2 = 1;
                }
            }
        }
    }

    /* ---- MiniGridEnv.step (G:1333-1441) */
    e->step_count += 1;
    int terminated = 0, truncated = 0;
    {   /* front_pos / fwd_cell are evaluated for every action (G:1341-1344): bounds assert */
        static const int DX[4] = { 1, 0, -1, 0 }, DY[4] = { 0, 1, 0, -1 };
        if (!in_bounds(e->ax + DX[e->dir], e->ay + DY[e->dir])) { e->error = E_ASSERT; return e->error; }
    }
    int tx = e->ax, ty = e->ay;
    switch (action) {
    case 0: tx -= 1; break;   /* left  G:1347 */
    case 1: tx += 1; break;   /* right G:1356 */
    case 2: ty -= 1; break;   /* up    G:1366 */
    case 3: ty += 1; break;   /* down  G:1376 */
    case 6: break;            /* done == stay G:1386 */
    default: e->error = E_ATTRIBUTE; return e->error;  /* self.actions.forward, G:1397 */
    }
    if (!in_bounds(tx, ty)) { e->error = E_ASSERT; return e->error; }
    {
        int k = ty * GS + tx, ct = e->type[k];
        if (ct == T_EMPTY || can_overlap(ct, e->state[k])) { e->ax = tx; e->ay = ty; }
        if (ct == T_GOAL) terminated = 1;               /* base reward is discarded, V6:181 */
    }
    if (e->step_count >= e->max_steps) truncated = 1;   /* G:1436-1437 */
    if (obs) tw_oracle_gen_obs(e, V, obs);              /* G:1439 */

    /* ---- Twoarmy post-logic */
    int reward = R_STEP;                                /* V6:181 */
    if (!e->pone && (e->ax > 3 || e->ay < 14)) {        /* V6:182-198 / V4:181-195 */
        int i1 = 11, i2 = 8;
        if (e->variant == 4) {
            i1 = 9 + (int)(take_draw(draws, seed, env_id, t, S_WALL1) % 4u);
            i2 = 6 + (int)(take_draw(draws, seed, env_id, t, S_WALL2) % 4u);
        }
        grid_set(e, 4, i1, T_WALL, C_GREY); grid_set(e, 5, i1, T_WALL, C_GREY);
        grid_set(e, 4, i1 + 1, T_WALL, C_GREY); grid_set(e, 5, i1 + 1, T_WALL, C_GREY);
        grid_set(e, i2, 11, T_WALL, C_GREY); grid_set(e, i2, 12, T_WALL, C_GREY);
        grid_set(e, i2 + 1, 11, T_WALL, C_GREY); grid_set(e, i2 + 1, 12, T_WALL, C_GREY);
        e->pone = 1;
    }
    if (e->variant == 4 && !e->patrol && e->ay <= 8) {  /* V4:212-225 */
        int i = 6 + (int)(take_draw(draws, seed, env_id, t, S_SPAWN) % 4u);
        const int px[4] = { i, i + 1, i, i + 1 }, py[4] = { 4, 4, 5, 5 };
        for (int k = 0; k < 4; ++k) { grid_set(e, px[k], py[k], T_BALL, C_YELLOW); e->o2_x[k] = px[k]; e->o2_y[k] = py[k]; }
        e->o2_valid = 1;
        for (int k = 0; k < 3; ++k) { grid_set(e, 12, k + 4, T_BALL, C_YELLOW); e->o1_x[k] = 12; e->o1_y[k] = k + 4; }
        e->o1_valid = 1;                                /* choice(range(4,5)) == 4, consumes nothing */
        e->patrol = 1;
    }
    for (int k = 0; k < 3; ++k)                         /* V6:231-239 */
        if (e->ax == e->ob_x[k] && e->ay == e->ob_y[k]) { reward = R_HIT; truncated = 1; }
    if (e->ay == e->ob_y[0] + 1 &&                      /* V6:241-243 */
        (e->ax == e->ob_x[0] || e->ax == e->ob_x[1] || e->ax == e->ob_x[2])) reward = R_RISK;
    if (e->patrol) {                                    /* V6:245-283 / V4:242-280 */
        if (!e->o1_valid || !e->o2_valid) { e->error = E_TYPE; return e->error; }
        if (e->ay == e->o2_y[2] + 1 && (e->ax == e->o2_x[2] || e->ax == e->o2_x[3])) reward = R_RISK;
        if (e->ax == e->o2_x[0] - 1 && (e->ay == e->o2_y[0] || e->ay == e->o2_y[2])) reward = R_RISK;
        if (e->ax == e->o2_x[1] + 1 && (e->ay == e->o2_y[1] || e->ay == e->o2_y[3])) reward = R_RISK;
        if (e->ax == e->o1_x[0] - 1 &&
            (e->ay == e->o1_y[0] || e->ay == e->o1_y[1] || e->ay == e->o1_y[2])) reward = R_RISK;
        for (int k = 0; k < 3; ++k)
            if (e->ax == e->o1_x[k] && e->ay == e->o1_y[k]) { reward = R_HIT; truncated = 1; }
        for (int k = 0; k < 4; ++k)
            if (e->ax == e->o2_x[k] && e->ay == e->o2_y[k]) { reward = R_HIT; truncated = 1; }
    }
    if (e->first_to_room2 && e->ay == 7) { reward = R_ROOM2; e->first_to_room2 = 0; }  /* V6:285-288 */
    if (reward == R_RISK) {                             /* V6:290-294 */
        e->risk_count += 1;
        if (e->risk_count > 5) truncated = 1;           /* `self.riskcount = 0` is a typo, no effect */
    }
    if (terminated || truncated) {                      /* V6:296-318 */
        if (terminated) reward = R_GOAL;
        e->step_move = 0; e->pone = 0; e->patrol = 0; e->first_to_room2 = 1; e->risk_count = 0;
        if ((take_draw(draws, seed, env_id, t, S_COIN_A) % 2u) == 1u) { e->up1 = 0; e->right2 = 1; }
        else { e->up1 = 1; e->right2 = 0; }
        if ((take_draw(draws, seed, env_id, t, S_COIN_B) % 2u) == 1u) { e->upd_horiz = 0; e->upd_long = 1; }
        else { e->upd_horiz = 1; e->upd_long = 0; }
    }
    e->reward_code = reward; e->terminated = terminated; e->truncated = truncated;
    return E_OK;
}

double tw_oracle_reward_value(int code) { return REWARD_VALUE[code]; }
int tw_oracle_sizeof_env(void) { return (int)sizeof(tw_env); }

/* ------------------------------------------------------------------ batched rollout
 * N independent envs (ids env0 .. env0+N-1), T steps each, with the training loop's
 * "reset after a done step" (soa/train_ppo.py:104,126,154).  Policy indices 0..4 are mapped
 * 4 -> 6 by Env_transact.env_action (EB:364-376).  actions == NULL -> Philox(seed, id, t, S_ACTION) % 5.
 * Any output pointer may be NULL.  Layout: [T][N][...].
 * `envs` (nullable): caller-provided state array (continues from it); else fresh envs.
 */
void tw_oracle_rollout(int variant, int N, int T, uint64_t seed, uint32_t env0, int V,
                       const int32_t *actions, tw_env *envs,
                       uint8_t *obs, float *matrix, float *pos, float *reward,
                       uint8_t *terminated, uint8_t *truncated, int autoreset) {
    tw_env *own = NULL;
    if (!envs) {
        own = (tw_env *)malloc(sizeof(tw_env) * (size_t)N);
        for (int n = 0; n < N; ++n) tw_oracle_init(&own[n], variant);
        envs = own;
    }
    const size_t osz = (size_t)V * V * 3;
    for (int n = 0; n < N; ++n) {
        tw_env *e = &envs[n];
        for (int t = 0; t < T; ++t) {
            size_t idx = (size_t)t * N + n;
            int a;
            if (actions) a = actions[idx];
            else { a = (int)(tw_oracle_draw_word(seed, env0 + n, e->t, S_ACTION) % 5u); }
            if (a == 4) a = 6;
            tw_oracle_step(e, a, NULL, seed, env0 + n, V, obs ? obs + idx * osz : NULL);
            if (matrix) tw_oracle_matrix(e, matrix + idx * NC);
            if (pos) { float g[2]; tw_oracle_pos(e, pos + idx * 2, g); }
            if (reward) reward[idx] = (float)REWARD_VALUE[e->reward_code];
            if (terminated) terminated[idx] = (uint8_t)e->terminated;
            if (truncated) truncated[idx] = (uint8_t)e->truncated;
            if (autoreset && (e->terminated || e->truncated)) tw_oracle_reset(e);
        }
    }
    free(own);
}
