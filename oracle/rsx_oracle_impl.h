/*
 * rsx_oracle_impl.h — body of the CPU oracle, instantiated for R = float and R = double by
 * rsx_oracle.c.  TEST INFRASTRUCTURE (see rsx_oracle.c header).  One env at a time, plain
 * loops; the ORDER of floating-point operations below is the specification that the HIP
 * kernels reproduce bit-for-bit in the float instantiation.
 *
 * State vector (wire format of robosim.get_state(), Entities/Frame.py:20-47 / :55-92):
 *   [0..4]  ball x, y, z, vx, vy         robot k at 5 + RS*k: x, y, theta(deg), vx, vy,
 *   omega(deg/s) [, infrared, v_wheel0..3 (rad/s)]   RS = 6 (VSS) | 11 (SSL)
 *   [state_dim] ball vertical velocity, [state_dim + 1] ball spin about the vertical axis in
 *   rad/s (both internal: not part of get_state(), carried by get/set_state_full).
 */

#define RC(x) ((R)(x))
#define MAXROB 22
#define MAXBOD 23

typedef struct SUF(rsxo_env) {
    rsxo_cfg cfg;
    int RS, state_dim;
    /* typed constants */
    R h, half_len, half_wid, ghw, gd, margin, r_robot, r_ball;
    R rs_rr, rs_rr2, rs_rb, rs_rb2;
    R w_rr, w_rb_r, w_rb_b, ope_rr, ope_rb, e_wb, e_wr, beta;
    int wall_aware;   /* model v2, per-axis shares of robot - robot pairs at a wall: 1 = SSL (probes, third / fourth sweep), 2 = VSS (held axes) — DESIGN.md 4 */
    R r_held;         /* VSS: a robot within 1 mm of where the wall clamp would hold it counts as held by that wall */
    R w_max, half_rw, rw_2b, inv_rw, r_wheel;
    R a_lin_h, a_lin_h2, a_lat_h, a_ang_h, mu_g_dt, g_h, e_ground, vz_min, robot_h;
    R dck_rb, half_kw, ir_tol, drib_gain, drib_vmax, drib_vmax2;
    /* tangential friction / ball spin */
    R dck, mu_rr, mu_rb, mu_wb, kt_rr, kt_rb_r, kt_rb_b, kw, spin_c, ope_wb, spin_dec_dt, pen2;
    R ws[4], wc[4], pinv[3][4];
    R deg2rad, rad2deg, h_deg;
    R state[5 + 11 * MAXROB + RSXO_XROWS];
    /* ---- task ---- */
    int task, obs_dim, act_dim, info_dim, max_steps;
    uint32_t key[2], env_id, episode;
    uint32_t tick;   /* step() calls since attach: what the per-step draws are keyed by */
    int steps;
    R max_pos, inv_max_pos, max_v, inv_max_v, inv_max_w, deadzone;
    R ou[MAXROB][2];
    R prev_pot, ep_ret;
    R info[10];
    R obs[64], final_obs[64], reward;
    R last_cmds[MAXROB * 8];
    uint8_t terminated, truncated;
    int64_t metrics[8];
    /* task constants */
    R hl_goal, inv_len_cm, inv_dt;
    R pen_x, half_pen_wid, inv_bd_scale, inv_bg_scale, inv_en_scale;
    R pl_xlo, pl_xspan, pl_ylo, pl_yspan, pl_min_d2;
    R ou_theta_dt, ou_sig_sqdt;
    R sc_sx, sc_sy, sc_j, sc_jb;   /* scrimmage line-up: grid spacing, robot jitter, ball jitter */
} SUF(rsxo_env);

static inline R SUF(clampr)(R v, R lo, R hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------ */
void* SUF(rsxo_create)(int kind, int field_type, int nb, int ny, int ts_ms) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)calloc(1, sizeof(SUF(rsxo_env)));
    if (!e) return NULL;
    if (rsxo_cfg_init(&e->cfg, kind, field_type, nb, ny, ts_ms)) { free(e); return NULL; }
    const rsxo_cfg* c = &e->cfg;
    e->RS = kind == 0 ? 6 : 11;
    e->state_dim = 5 + e->RS * c->n_robots;
    e->h = RC(c->h);
    e->half_len = RC(c->half_len); e->half_wid = RC(c->half_wid);
    e->ghw = RC(c->goal_half_wid); e->gd = RC(c->goal_depth); e->margin = RC(c->margin);
    e->r_robot = RC(c->r_robot); e->r_ball = RC(c->r_ball);
    e->rs_rr = RC(2.0 * c->r_robot); e->rs_rr2 = RC((2.0 * c->r_robot) * (2.0 * c->r_robot));
    e->rs_rb = RC(c->r_robot + c->r_ball);
    e->rs_rb2 = RC((c->r_robot + c->r_ball) * (c->r_robot + c->r_ball));
    double imr = 1.0 / c->m_robot, imb = 1.0 / c->m_ball;
    e->wall_aware = c->kind == 1 ? 1 : 2;
    e->r_held = RC(c->r_robot + 0.001);
    e->w_rr = RC(0.5); e->w_rb_r = RC(imr / (imr + imb)); e->w_rb_b = RC(imb / (imr + imb));
    e->ope_rr = RC(1.0 + c->e_rr); e->ope_rb = RC(1.0 + c->e_rb);
    e->e_wb = RC(c->e_wall_ball); e->e_wr = RC(c->e_wall_robot); e->beta = RC(c->beta);
    e->w_max = RC(c->w_max); e->r_wheel = RC(c->r_wheel);
    e->half_rw = RC(c->r_wheel * 0.5); e->rw_2b = RC(c->r_wheel / (2.0 * c->lever));
    e->inv_rw = RC(1.0 / c->r_wheel);
    e->a_lin_h = RC(c->a_lin * c->h); e->a_lin_h2 = RC((c->a_lin * c->h) * (c->a_lin * c->h));
    e->a_lat_h = RC(c->a_lat * c->h); e->a_ang_h = RC(c->a_ang * c->h);
    e->mu_g_dt = RC(c->mu_g * (c->time_step_ms * 0.001)); e->g_h = RC(c->grav * c->h);
    e->e_ground = RC(c->e_ground); e->vz_min = RC(c->vz_min); e->robot_h = RC(c->robot_h);
    e->dck_rb = RC(c->dck + c->r_ball); e->half_kw = RC(c->half_kw); e->ir_tol = RC(c->ir_tol);
    e->drib_gain = RC(c->h > 0 ? 0.5 / c->h : 0.0);
    e->drib_vmax = RC(c->drib_vmax); e->drib_vmax2 = RC(c->drib_vmax * c->drib_vmax);
    /* Coulomb friction in contacts.  Tangential effective mass: robots are yaw-controlled by
     * their motors (no torque from contacts), the ball is a solid sphere (I = 2/5 m r^2, so the
     * contact point adds r^2 / I = 2.5 / m):  1/m_t = 1/m_r + 3.5/m_b (robot-ball), 2/m_r
     * (robot-robot), 3.5/m_b (wall-ball). */
    {
        double mt_rb = 1.0 / (imr + 3.5 * imb);
        e->dck = RC(c->dck);
        e->mu_rr = RC(c->mu_rr); e->mu_rb = RC(c->mu_rb); e->mu_wb = RC(c->mu_wb);
        e->kt_rr = RC(0.5); e->kt_rb_r = RC(mt_rb * imr); e->kt_rb_b = RC(mt_rb * imb);
        e->kw = RC(2.0 / 7.0); e->spin_c = RC(2.5 / c->r_ball);
        e->ope_wb = RC(1.0 + c->e_wall_ball);
        e->spin_dec_dt = RC(c->spin_dec * (c->time_step_ms * 0.001));
        e->pen2 = RC(c->pen2);
    }
    for (int k = 0; k < 4; ++k) { e->ws[k] = RC(sin(c->wheel_ang[k])); e->wc[k] = RC(cos(c->wheel_ang[k])); }
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 4; ++k) e->pinv[i][k] = RC(c->pinv[i][k]);
    e->deg2rad = RC(RSXO_PI / 180.0); e->rad2deg = RC(180.0 / RSXO_PI);
    e->h_deg = RC(c->h * (180.0 / RSXO_PI)); /* heading is integrated in degrees, the wire unit */
    /* adapter's dummy line-up, rsim.py:20-24 */
    e->state[2] = e->r_ball;
    for (int k = 0; k < c->n_robots; ++k) {
        int i = k < nb ? k + 1 : k - nb + 1;
        e->state[5 + e->RS * k] = RC((k < nb ? -0.2 : 0.2) * i);
    }
    return e;
}
void SUF(rsxo_destroy)(void* p) { free(p); }
int SUF(rsxo_state_dim)(void* p) { return ((SUF(rsxo_env)*)p)->state_dim; }

void SUF(rsxo_field_params)(void* p, double out[17]) {
    memcpy(out, ((SUF(rsxo_env)*)p)->cfg.field, 17 * sizeof(double));
}

/* robosim.reset(ball, blue, yellow) — rsim.py:38,52-75 */
void SUF(rsxo_reset)(void* p, const double* ball, const double* blue, const double* yellow) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    const rsxo_cfg* c = &e->cfg;
    R* s = e->state;
    memset(s, 0, sizeof(e->state));
    s[0] = RC(ball[0]); s[1] = RC(ball[1]); s[2] = e->r_ball; s[3] = RC(ball[2]); s[4] = RC(ball[3]);
    for (int k = 0; k < c->n_robots; ++k) {
        const double* src = k < c->n_blue ? blue + 3 * k : yellow + 3 * (k - c->n_blue);
        R* r = s + 5 + e->RS * k;
        r[0] = RC(src[0]); r[1] = RC(src[1]); r[2] = RC(src[2]);
    }
}
void SUF(rsxo_get_state)(void* p, double* out) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    for (int i = 0; i < e->state_dim; ++i) out[i] = (double)e->state[i];
}
void SUF(rsxo_get_state_full)(void* p, double* out) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    for (int i = 0; i < e->state_dim + RSXO_XROWS; ++i) out[i] = (double)e->state[i];
}
void SUF(rsxo_set_state_full)(void* p, const double* in) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    for (int i = 0; i < e->state_dim + RSXO_XROWS; ++i) e->state[i] = RC(in[i]);
}

/* ------------------------------------------------------------------------------------------
 * walls: clamp a circle of radius r into the playable region; e = restitution
 * ---------------------------------------------------------------------------------------- */
static int SUF(walls)(const SUF(rsxo_env)* e, R r, R rest, R* px, R* py, R* pvx, R* pvy) {
    R x = *px, y = *py, vx = *pvx, vy = *pvy;
    R ax = R_FABS(x), ay = R_FABS(y);
    R sx = x < RC(0) ? RC(-1) : RC(1), sy = y < RC(0) ? RC(-1) : RC(1);
    int hit = 0; /* bit 0: vx was reflected, bit 1: vy was reflected */
    if (e->cfg.kind == 0) {
        if (ax > e->half_len) { /* centre inside a goal box */
            R yl = e->ghw - r;
            if (ay > yl) { y = sy * yl; if (vy * sy > RC(0)) { vy = -rest * vy; hit |= 2; } }
            R xl = (e->half_len + e->gd) - r;
            if (ax > xl) { x = sx * xl; if (vx * sx > RC(0)) { vx = -rest * vx; hit |= 1; } }
        } else {
            R yl = e->half_wid - r;
            if (ay > yl) { y = sy * yl; if (vy * sy > RC(0)) { vy = -rest * vy; hit |= 2; } ay = yl; }
            /* the goal line's wall exists beside the goal mouth only (|y| >= goal half width) and ends in a goal post (model v2): a
             * body keeps its radius from the corner (+-L/2, +-goal half width) — along the CHORD of that arc: with u, v the centre's
             * distances to the goal line and to the mouth's edge (both positive in the corner region), u + v >= r, i.e.
             * |x| + |y| <= (L/2 + goal half width) - r.  The chord joins the goal line's limit (u = r at v = 0) to the goal's side-wall
             * limit (v = r at u = 0), so the limits are continuous all the way round; a body cuts the corner by at most 0.29 r.
             * (v1 clamped x to the goal line's limit for |y| > goal half width - r: a body that slid along the mouth's edge into that
             * strip was thrown up to a radius sideways, into whatever stood there — the 3 cm overlaps of the VSS scrum.)  Pushed out
             * along the chord's normal (1, 1) / sqrt 2, folded into the first quadrant; the velocity component along it is reflected. */
            R xl = e->half_len - r;
            if (ay >= e->ghw) {
                if (ax > xl) { x = sx * xl; if (vx * sx > RC(0)) { vx = -rest * vx; hit |= 1; } }
            } else {
                R cl = (e->half_len + e->ghw) - r, sa = ax + ay;
                if (sa > cl) {
                    R hf = RC(0.5) * (sa - cl);
                    x = sx * (ax - hf); y = sy * (ay - hf);
                    R vn = R_FMA(vx, sx, vy * sy);     /* sqrt 2 x the speed along the normal; > 0: moving into the post */
                    if (vn > RC(0)) {
                        R dv = ((RC(1) + rest) * RC(0.5)) * vn;
                        vx = R_FMA(-sx, dv, vx); vy = R_FMA(-sy, dv, vy);
                        hit |= 4;
                    }
                }
            }
        }
    } else {
        R yl = (e->half_wid + e->margin) - r;
        if (ay > yl) { y = sy * yl; if (vy * sy > RC(0)) { vy = -rest * vy; hit |= 2; } ay = yl; }
        R xl = (e->half_len + e->margin) - r;
        if (ax > xl) { x = sx * xl; if (vx * sx > RC(0)) { vx = -rest * vx; hit |= 1; } ax = xl; }
        {   /* goal posts: the open ends of the goal's side walls, points at (+-L/2, +-goal_width/2) that a body keeps its radius
             * from (without them a body overlaps the wall's end from the field side and is thrown sideways by the side-wall clamp
             * the moment it crosses the goal line).  Folded into the first quadrant; n points post -> body. */
            R dxp = ax - e->half_len, dyp = ay - e->ghw;
            R d2 = R_FMA(dxp, dxp, dyp * dyp);
            if (d2 < r * r && d2 > RC(0)) {
                R d = R_SQRT(d2), inv = RC(1) / d;
                R nxp = dxp * inv, nyp = dyp * inv;
                ax = R_FMA(r, nxp, e->half_len); ay = R_FMA(r, nyp, e->ghw);
                x = sx * ax; y = sy * ay;
                R vr = R_FMA(vx * sx, nxp, (vy * sy) * nyp);     /* radial speed; < 0: moving into the post */
                if (vr < RC(0)) {
                    R dv = -((RC(1) + rest) * vr);
                    vx = R_FMA(sx, dv * nxp, vx); vy = R_FMA(sy, dv * nyp, vy);
                    hit |= 4;
                }
            }
        }
        if (ax > e->half_len) {
            R back = e->half_len + e->gd;
            if (ay < e->ghw) {
                if (ax < back) { /* inside the goal */
                    if (ax > back - r) { x = sx * (back - r); if (vx * sx > RC(0)) { vx = -rest * vx; hit |= 1; } }
                    if (ay > e->ghw - r) { y = sy * (e->ghw - r); if (vy * sy > RC(0)) { vy = -rest * vy; hit |= 2; } }
                } else if (ax < back + r) { /* behind the back wall */
                    x = sx * (back + r); if (vx * sx < RC(0)) { vx = -rest * vx; hit |= 1; }
                }
            } else if (ay < e->ghw + r && ax < back) { /* outside, touching a side wall */
                y = sy * (e->ghw + r); if (vy * sy < RC(0)) { vy = -rest * vy; hit |= 2; }
            }
        }
    }
    *px = x; *py = y; *pvx = vx; *pvy = vy;
    return hit;
}

/* A bounce of the BALL off a wall with Coulomb friction at the contact point: couples the
 * velocity component along the wall with the spin about the vertical axis.  (vx0, vy0) is the
 * velocity before walls(): the ball moved INTO the wall, so its sign names the wall's side.
 * Wall along x (vy reflected) first, then wall along y. */
static inline void SUF(ball_wall_spin)(const SUF(rsxo_env)* e, int hit, R vx0, R vy0, R* vx, R* vy, R* om) {
    if (hit & 2) {
        R sg = vy0 < RC(0) ? RC(-1) : RC(1);
        R vc = *vx - (*om * e->r_ball) * sg;                 /* contact-point speed along the wall */
        R lim = e->mu_wb * (e->ope_wb * R_FABS(vy0));
        R d = SUF(clampr)(-(vc * e->kw), -lim, lim);
        *vx = *vx + d; *om = *om - (sg * d) * e->spin_c;
    }
    if (hit & 1) {
        R sg = vx0 < RC(0) ? RC(-1) : RC(1);
        R vc = *vy + (*om * e->r_ball) * sg;
        R lim = e->mu_wb * (e->ope_wb * R_FABS(vx0));
        R d = SUF(clampr)(-(vc * e->kw), -lim, lim);
        *vy = *vy + d; *om = *om + (sg * d) * e->spin_c;
    }
}

/* heading after a small turn d (rad): rotate (c, s) by sin / cos of d (|d| < 0.5; odd / even
 * Taylor polynomials, error < 1e-8) — one exact sincos per step(), cheap rotations per sub-step */
static inline void SUF(rotate_heading)(R d, R* c, R* s) {
    R d2 = d * d;
    R sd = d * R_FMA(d2, R_FMA(d2, RC(8.3333333333333333e-3), RC(-1.6666666666666667e-1)), RC(1));
    R cd = R_FMA(d2, R_FMA(d2, R_FMA(d2, RC(-1.3888888888888889e-3), RC(4.1666666666666664e-2)), RC(-0.5)), RC(1));
    R c0 = *c, s0 = *s;
    *c = R_FMA(c0, cd, -(s0 * sd));
    *s = R_FMA(s0, cd, c0 * sd);
}

/* per-body working record */
typedef struct SUF(body) {
    R x, y, vx, vy;        /* all */
    R th, om, c, s;        /* robots: heading (DEGREES), rate (rad/s), cos/sin(heading); ball: om = spin (rad/s) */
    R t0, t1, t2;          /* VSS: (v_target, om_target, -) | SSL: (vtx, vty, om_target) */
    R kick_x, kick_z; int drib, ir;
    R z, vz;               /* ball */
} SUF(body);

/* robot(a) - ball(b) contact geometry: normal n (a -> b), penetration, mouth flag */
static inline int SUF(rb_geom)(const SUF(rsxo_env)* e, const SUF(body)* a, const SUF(body)* b,
                               R* nx, R* ny, R* pen, int* mouth) {
    R dx = b->x - a->x, dy = b->y - a->y;
    *mouth = 0;
    if (b->z >= e->robot_h) { *pen = RC(-1); *nx = RC(0); *ny = RC(0); return 0; }
    if (e->cfg.kind == 1) {
        R lx = R_FMA(dx, a->c, dy * a->s), ly = R_FMA(dy, a->c, -(dx * a->s));
        if (R_FABS(ly) < e->half_kw && lx > RC(0)) {
            *mouth = 1; *pen = e->dck_rb - lx; *nx = a->c; *ny = a->s;
            return *pen > RC(0);
        }
    }
    R d2 = R_FMA(dx, dx, dy * dy);
    if (d2 < e->rs_rb2 && d2 > RC(0)) {
        R d = R_SQRT(d2), inv = RC(1) / d;
        *nx = dx * inv; *ny = dy * inv; *pen = e->rs_rb - d;
        return 1;
    }
    *pen = RC(-1); *nx = RC(0); *ny = RC(0);
    return 0;
}

/* Response of body a to ONE touching partner p, from a's point of view (each side of a pair
 * evaluates this with its own constants).  n = unit normal a -> p, pen = penetration depth,
 * (dvx, dvy) = v_p - v_a, wsum = om_p * lever_p + om_a * lever_a (surface speeds at the contact
 * point), w = a's share of the normal impulse, kt = a's share of the tangential one, mu = Coulomb
 * coefficient, spin_c = spin gained per unit of tangential velocity change (ball only). */
static inline void SUF(respond)(const SUF(rsxo_env)* e, R nx, R ny, R pen, R dvx, R dvy, R wsum,
                                R ope, R w, R wx, R wy, R kt, R mu, R spin_c,
                                R* avx, R* avy, R* apx, R* apy, R* aw) {
    /* (wx, wy): a's share of the normal impulse and of the de-penetration, per axis — w on both axes except for a robot - robot
     * pair at a wall (wall_shares below); the Coulomb limit keeps the nominal share w */
    R vn = R_FMA(dvx, nx, dvy * ny);
    if (vn < RC(0)) {
        R q = ope * vn;                                   /* <= 0: pushes a away from p */
        *avx = R_FMA(q * wx, nx, *avx); *avy = R_FMA(q * wy, ny, *avy);
        R vt = R_FMA(dvy, nx, -(dvx * ny)) - wsum;        /* along t = (-ny, nx) */
        R lim = (q * w) * mu;
        R ft = SUF(clampr)(vt * kt, lim, -lim);           /* sticking impulse, Coulomb-limited */
        *avx = R_FMA(-ft, ny, *avx); *avy = R_FMA(ft, nx, *avy);
        *aw = R_FMA(ft, spin_c, *aw);
    }
    R pc = e->beta * pen;
    *apx = R_FMA(-(pc * wx), nx, *apx); *apy = R_FMA(-(pc * wy), ny, *apy);
}

/* Model v2 (SSL): a robot that stands against a wall cannot yield along that wall's normal: in a robot - robot pair the partner
 * then takes the whole correction on that axis (the wall holds the other side).  Without this a pile that the robots' own push
 * presses against a wall overlaps by centimetres: the contact phase moves the outer robot into the wall, the wall clamp puts it back
 * into its neighbour.  Blocked axes are found by PROBING: each body displaced by 1 mm the way this contact pushes it (a: against n,
 * p: along n); an axis is blocked when the probe lies where the wall clamp (walls above, a robot's radius) acts on that coordinate:
 *   x: beyond the end wall's limit | within r of a goal post | in a goal: at its back wall from inside, or behind it
 *   y: beyond the side wall's limit | within r of a goal post | in a goal: at a side wall from inside, or beside it from outside
 * Shares per axis: blocked body 0, its free partner 1, otherwise 1/2 each.  Returns whether any axis of either body was blocked
 * (a "wall pair"). */
static inline void SUF(blocked_axes)(const SUF(rsxo_env)* e, R x, R y, int* bx, int* by) {
    const R r = e->r_robot;
    const R ax = R_FABS(x), ay = R_FABS(y);
    const R dxp = ax - e->half_len, dyp = ay - e->ghw;
    const R d2 = R_FMA(dxp, dxp, dyp * dyp);
    const int post = d2 < r * r && d2 > RC(0);
    const R back = e->half_len + e->gd;
    const int beyond = ax > e->half_len, in_mouth = ay < e->ghw, inside = in_mouth && ax < back;
    *bx = ax > (e->half_len + e->margin) - r || post || (beyond && ((inside && ax > back - r) || (in_mouth && !(ax < back) && ax < back + r)));
    *by = ay > (e->half_wid + e->margin) - r || post || (beyond && ((inside && ay > e->ghw - r) || (!in_mouth && ay < e->ghw + r && ax < back)));
}
static inline int SUF(wall_shares)(const SUF(rsxo_env)* e, R xa, R ya, R xp, R yp, R nx, R ny, R* wx, R* wy) {
    const R eps = RC(0.001);
    int abx, aby, pbx, pby;
    SUF(blocked_axes)(e, xa - eps * nx, ya - eps * ny, &abx, &aby);
    SUF(blocked_axes)(e, xp + eps * nx, yp + eps * ny, &pbx, &pby);
    *wx = (abx && !pbx) ? RC(0) : ((!abx && pbx) ? RC(1) : RC(0.5));
    *wy = (aby && !pby) ? RC(0) : ((!aby && pby) ? RC(1) : RC(0.5));
    return abx | aby | pbx | pby;
}

/* Model v2 (VSS): the same idea without probes or directions.  A robot is HELD on an axis when its centre lies within 1 mm of where
 * the wall clamp (walls above, a robot's radius) limits that coordinate:
 *   y: |y| >= limit - 1 mm, the limit being the touch line's (field) or the goal's side wall's (inside a goal box)
 *   x: |x| >= limit - 1 mm, the limit being the goal line's wall (beside the goal mouth only) or the goal's back wall (inside a goal box)
 * (goal posts hold nothing: a robot slides around them).  h = 1 when held, 0 otherwise.  Shares per axis of a robot - robot pair
 * (a: this body, p: its partner):  1/2 + (h_p - h_a) / 2  ->  held body 0, its free partner 1, otherwise 1/2 each.  (Which way the
 * contact pushes is not asked: a partner that could push a held robot AWAY from its wall would have to stand between the robot and
 * the wall, i.e. be held itself — measured: the scrum envelope is the same with and without a direction test.)
 * Evaluated on the snapshot the sweep reads.  VSS stays at two sweeps (measured sufficient: profiles/r06_jam_vss_model_v2.txt). */
static inline void SUF(held_axes)(const SUF(rsxo_env)* e, R x, R y, R* hx, R* hy) {
    const R ax = R_FABS(x), ay = R_FABS(y);
    const int in_goal = ax > e->half_len;
    const R yl = (in_goal ? e->ghw : e->half_wid) - e->r_held;
    const R xl = (in_goal ? e->half_len + e->gd : e->half_len) - e->r_held;
    *hy = ay >= yl ? RC(1) : RC(0);
    *hx = (ax >= xl && (in_goal || ay >= e->ghw)) ? RC(1) : RC(0);
}
static inline void SUF(held_shares)(const SUF(rsxo_env)* e, R xa, R ya, R xp, R yp, R* wx, R* wy) {
    R hax, hay, hpx, hpy;
    SUF(held_axes)(e, xa, ya, &hax, &hay);
    SUF(held_axes)(e, xp, yp, &hpx, &hpy);
    *wx = R_FMA(hpx - hax, RC(0.5), RC(0.5));
    *wy = R_FMA(hpy - hay, RC(0.5), RC(0.5));
}

typedef struct SUF(kick) { int ovr, okick; R ovx, ovy, ovz; } SUF(kick);

/* One Jacobi sweep: every body sums the responses to its touching partners (index order) from
 * the SAME snapshot, then all are applied.  Returns 1 when some pair overlapped by more than
 * pen2 (a deep contact: an impact at speed or a jammed pile — resting contacts stay far below).
 * first != 0: infrared sensors are refreshed and kicker / dribbler decisions recorded in K. */
static int SUF(contacts)(SUF(rsxo_env)* e, SUF(body)* b, int first, SUF(kick)* K) {
    const rsxo_cfg* c = &e->cfg;
    const int N = c->n_robots, M = N + 1, ssl = c->kind == 1;
    SUF(body)* ball = &b[N];
    R dvx[MAXBOD], dvy[MAXBOD], dpx[MAXBOD], dpy[MAXBOD], dws = RC(0);
    int got[MAXBOD];   /* body had at least one touching partner: only those are updated */
    int any = 0, wallpair = 0;
    for (int i = 0; i < M; ++i) {
        R avx = RC(0), avy = RC(0), apx = RC(0), apy = RC(0), aw = RC(0);
        int touched = 0;
        for (int j = 0; j < M; ++j) {
            if (j == i) continue;
            if (i < N && j < N) { /* robot - robot */
                R dx = b[j].x - b[i].x, dy = b[j].y - b[i].y;
                R d2 = R_FMA(dx, dx, dy * dy);
                if (d2 < e->rs_rr2 && d2 > RC(0)) {
                    R d = R_SQRT(d2), inv = RC(1) / d;
                    R wsum = R_FMA(b[j].om, e->r_robot, b[i].om * e->r_robot);
                    R wx = e->w_rr, wy = e->w_rr;
                    if (e->wall_aware == 1 && SUF(wall_shares)(e, b[i].x, b[i].y, b[j].x, b[j].y, dx * inv, dy * inv, &wx, &wy)) wallpair = 1;
                    if (e->wall_aware == 2) SUF(held_shares)(e, b[i].x, b[i].y, b[j].x, b[j].y, &wx, &wy);
                    SUF(respond)(e, dx * inv, dy * inv, e->rs_rr - d, b[j].vx - b[i].vx, b[j].vy - b[i].vy, wsum,
                                 e->ope_rr, e->w_rr, wx, wy, e->kt_rr, e->mu_rr, RC(0), &avx, &avy, &apx, &apy, &aw);
                    if (e->rs_rr - d > e->pen2) any = 1;
                    touched = 1;
                }
            } else if (i < N) { /* robot i, ball j */
                R nx, ny, pen; int mouth;
                int touch = SUF(rb_geom)(e, &b[i], ball, &nx, &ny, &pen, &mouth);
                if (touch) {
                    R wsum = R_FMA(ball->om, e->r_ball, b[i].om * (mouth ? e->dck : e->r_robot));
                    SUF(respond)(e, nx, ny, pen, ball->vx - b[i].vx, ball->vy - b[i].vy, wsum,
                                 e->ope_rb, e->w_rb_r, e->w_rb_r, e->w_rb_r, e->kt_rb_r, e->mu_rb, RC(0), &avx, &avy, &apx, &apy, &aw);
                    if (pen > e->pen2) any = 1;
                    touched = 1;
                }
                if (first) b[i].ir = mouth && pen > -e->ir_tol;
            } else if (!ssl) { /* VSS ball i, robot j: circle - circle from the ball's point of view */
                R dx = b[j].x - ball->x, dy = b[j].y - ball->y;
                R d2 = R_FMA(dx, dx, dy * dy);
                if (d2 < e->rs_rb2 && d2 > RC(0) && ball->z < e->robot_h) {
                    R d = R_SQRT(d2), inv = RC(1) / d;
                    R wsum = R_FMA(b[j].om, e->r_robot, ball->om * e->r_ball);
                    SUF(respond)(e, dx * inv, dy * inv, e->rs_rb - d, b[j].vx - ball->vx, b[j].vy - ball->vy, wsum,
                                 e->ope_rb, e->w_rb_b, e->w_rb_b, e->w_rb_b, e->kt_rb_b, e->mu_rb, e->spin_c, &avx, &avy, &apx, &apy, &aw);
                    if (e->rs_rb - d > e->pen2) any = 1;
                    touched = 1;
                }
            } else { /* SSL ball i, robot j: the ball's side of what robot j evaluated (robot frame) */
                R nx, ny, pen; int mouth;
                int touch = SUF(rb_geom)(e, &b[j], ball, &nx, &ny, &pen, &mouth);
                if (touch) {
                    R dx_ = ball->vx - b[j].vx, dy_ = ball->vy - b[j].vy;
                    R vn = R_FMA(dx_, nx, dy_ * ny);
                    if (vn < RC(0)) {
                        R q = e->ope_rb * vn * e->w_rb_b;
                        R wsum = R_FMA(ball->om, e->r_ball, b[j].om * (mouth ? e->dck : e->r_robot));
                        R vt = R_FMA(dy_, nx, -(dx_ * ny)) - wsum;
                        R lim = q * e->mu_rb;
                        R ft = SUF(clampr)(vt * e->kt_rb_b, lim, -lim);
                        avx = avx - R_FMA(-ft, ny, q * nx);
                        avy = avy - R_FMA(ft, nx, q * ny);
                        aw = aw + ft * e->spin_c;
                    }
                    R pc = e->beta * pen * e->w_rb_b;
                    apx = apx + pc * nx; apy = apy + pc * ny;
                    if (pen > e->pen2) any = 1;
                    touched = 1;
                }
                if (first && mouth && pen > -e->ir_tol) { /* infrared: kicker / dribbler act */
                    if (b[j].kick_x > RC(0) || b[j].kick_z > RC(0)) {
                        K->ovr = 1; K->okick = 1;
                        K->ovx = b[j].vx + b[j].kick_x * b[j].c;
                        K->ovy = b[j].vy + b[j].kick_x * b[j].s;
                        K->ovz = b[j].kick_z;
                    } else if (b[j].drib) {
                        R hx = b[j].x + e->dck_rb * b[j].c, hy = b[j].y + e->dck_rb * b[j].s;
                        R cvx = (hx - ball->x) * e->drib_gain, cvy = (hy - ball->y) * e->drib_gain;
                        R m2 = cvx * cvx + cvy * cvy;
                        if (m2 > e->drib_vmax2) { R sc = e->drib_vmax / R_SQRT(m2); cvx = cvx * sc; cvy = cvy * sc; }
                        K->ovr = 1; K->okick = 0;
                        K->ovx = (b[j].vx - b[j].om * e->dck_rb * b[j].s) + cvx;
                        K->ovy = (b[j].vy + b[j].om * e->dck_rb * b[j].c) + cvy;
                    }
                }
            }
        }
        dvx[i] = avx; dvy[i] = avy; dpx[i] = apx; dpy[i] = apy; got[i] = touched;
        if (i == N) dws = aw;
    }
    for (int i = 0; i < M; ++i) {
        if (!got[i]) continue;   /* bodies without a contact keep their bits */
        b[i].vx = b[i].vx + dvx[i]; b[i].vy = b[i].vy + dvy[i];
        b[i].x = b[i].x + dpx[i]; b[i].y = b[i].y + dpy[i];
    }
    if (got[N]) ball->om = ball->om + dws;
    return any | (wallpair && any ? 2 : 0);   /* bit 1: deep, and some touching robot - robot pair stood at a wall */
}

/* ------------------------------------------------------------------------------------------
 * robosim.step(cmds) — rsim.py:102 (VSS [N][2]) / rsim.py:155 (SSL [N][8]); cmds already R
 * ---------------------------------------------------------------------------------------- */
static void SUF(step_core)(SUF(rsxo_env)* e, const R* cmds) {
    const rsxo_cfg* c = &e->cfg;
    const int N = c->n_robots, RS = e->RS, ssl = c->kind == 1;
    SUF(body) b[MAXBOD];
    memset(b, 0, sizeof(b));
    R* s = e->state;
    /* ---- load + per-step command processing ---- */
    for (int k = 0; k < N; ++k) {
        const R* r = s + 5 + RS * k;
        SUF(body)* o = &b[k];
        o->x = r[0]; o->y = r[1]; o->th = r[2]; o->vx = r[3]; o->vy = r[4];
        o->om = r[5] * e->deg2rad;
        R_SINCOS(o->th * e->deg2rad, &o->s, &o->c);
        if (!ssl) {
            R wl = SUF(clampr)(cmds[2 * k], -e->w_max, e->w_max);
            R wr = SUF(clampr)(cmds[2 * k + 1], -e->w_max, e->w_max);
            o->t0 = (wl + wr) * e->half_rw;
            o->t1 = (wr - wl) * e->rw_2b;
        } else {
            const R* q = cmds + 8 * k;
            R vtx, vty, omt;
            if (q[0] != RC(0)) {
                R w[4];
                for (int i = 0; i < 4; ++i) w[i] = SUF(clampr)(q[1 + i], -e->w_max, e->w_max);
                vtx = (((e->pinv[0][0] * w[0] + e->pinv[0][1] * w[1]) + e->pinv[0][2] * w[2]) + e->pinv[0][3] * w[3]) * e->r_wheel;
                vty = (((e->pinv[1][0] * w[0] + e->pinv[1][1] * w[1]) + e->pinv[1][2] * w[2]) + e->pinv[1][3] * w[3]) * e->r_wheel;
                omt = (((e->pinv[2][0] * w[0] + e->pinv[2][1] * w[1]) + e->pinv[2][2] * w[2]) + e->pinv[2][3] * w[3]) * e->r_wheel;
            } else {
                vtx = q[1]; vty = q[2]; omt = q[3];
                R m = RC(0);
                for (int i = 0; i < 4; ++i) {
                    R wi = ((vty * e->wc[i] - vtx * e->ws[i]) + omt * e->r_robot) * e->inv_rw;
                    R a = R_FABS(wi);
                    if (a > m) m = a;
                }
                if (m > e->w_max) { R sc = e->w_max / m; vtx = vtx * sc; vty = vty * sc; omt = omt * sc; }
            }
            o->t0 = vtx; o->t1 = vty; o->t2 = omt;
            o->kick_x = q[5]; o->kick_z = q[6]; o->drib = q[7] != RC(0);
        }
    }
    SUF(body)* ball = &b[N];
    ball->x = s[0]; ball->y = s[1]; ball->z = s[2] - e->r_ball; ball->vx = s[3]; ball->vy = s[4];
    ball->vz = s[e->state_dim];
    ball->om = s[e->state_dim + 1];
    /* rolling resistance: a constant deceleration, applied once for the whole step() while the
     * ball is on the ground (exact stop, never reverses) */
    if (c->n_sub && !(ball->z > RC(0) || ball->vz > RC(0))) {
        R sp2 = R_FMA(ball->vx, ball->vx, ball->vy * ball->vy);
        if (sp2 > RC(0)) {
            R sp = R_SQRT(sp2), ns = sp - e->mu_g_dt;
            if (ns < RC(0)) ns = RC(0);
            R k = ns / sp;
            ball->vx = ball->vx * k; ball->vy = ball->vy * k;
        }
        /* spin about the vertical axis: constant pivoting-friction deceleration to an exact stop */
        R aw = R_FABS(ball->om) - e->spin_dec_dt;
        ball->om = aw > RC(0) ? (ball->om < RC(0) ? -aw : aw) : RC(0);
    }

    for (int sub = 0; sub < c->n_sub; ++sub) {
        /* ---- A: actuation + integration ---- */
        for (int k = 0; k < N; ++k) {
            SUF(body)* o = &b[k];
            R vf = R_FMA(o->vy, o->s, o->vx * o->c);
            R vl = R_FMA(o->vy, o->c, -(o->vx * o->s));
            if (!ssl) {
                vf = vf + SUF(clampr)(o->t0 - vf, -e->a_lin_h, e->a_lin_h);
                vl = vl - SUF(clampr)(vl, -e->a_lat_h, e->a_lat_h);
                o->om = o->om + SUF(clampr)(o->t1 - o->om, -e->a_ang_h, e->a_ang_h);
            } else {
                R dx = o->t0 - vf, dy = o->t1 - vl;
                R d2 = R_FMA(dx, dx, dy * dy);
                if (d2 > e->a_lin_h2) { R sc = e->a_lin_h / R_SQRT(d2); dx = dx * sc; dy = dy * sc; }
                vf = vf + dx; vl = vl + dy;
                o->om = o->om + SUF(clampr)(o->t2 - o->om, -e->a_ang_h, e->a_ang_h);
            }
            o->vx = R_FMA(vf, o->c, -(vl * o->s));
            o->vy = R_FMA(vf, o->s, vl * o->c);
            o->x = R_FMA(o->vx, e->h, o->x);
            o->y = R_FMA(o->vy, e->h, o->y);
            o->th = R_FMA(o->om, e->h_deg, o->th);
            if (o->th > RC(180)) o->th = o->th - RC(360);
            else if (o->th < RC(-180)) o->th = o->th + RC(360);
            SUF(rotate_heading)(o->om * e->h, &o->c, &o->s);
        }
        if (ball->z > RC(0) || ball->vz > RC(0)) {
            ball->vz = ball->vz - e->g_h;
            ball->z = R_FMA(ball->vz, e->h, ball->z);
            if (ball->z <= RC(0)) {
                ball->z = RC(0);
                ball->vz = -ball->vz * e->e_ground;
                if (ball->vz < e->vz_min) ball->vz = RC(0);
            }
        }
        ball->x = R_FMA(ball->vx, e->h, ball->x);
        ball->y = R_FMA(ball->vy, e->h, ball->y);

        /* ---- B: contacts — one Jacobi sweep over the post-integration snapshot, and a second
         * one over the corrected snapshot when some pair was deep (impacts at speed, jammed piles);
         * what kicker and dribbler decided in the first sweep is applied after the impulses ---- */
        SUF(kick) K; memset(&K, 0, sizeof(K));
        if (SUF(contacts)(e, b, 1, &K)) {
            int deep = SUF(contacts)(e, b, 0, &K);
            /* a third and a fourth sweep for piles at a wall: while the last sweep saw a deep pair AND a touching robot - robot pair
             * with a wall-blocked axis (the share-1 corrections of wall pairs need the extra relaxation; piles in the open do not) */
            for (int sw = 2; sw < 4 && (deep & 2); ++sw) deep = SUF(contacts)(e, b, 0, &K);
        }
        if (K.ovr) {
            ball->vx = K.ovx; ball->vy = K.ovy; ball->om = RC(0);
            if (K.okick && K.ovz > RC(0)) ball->vz = K.ovz;
        }
        /* ---- C: walls ---- */
        for (int k = 0; k < N; ++k) SUF(walls)(e, e->r_robot, e->e_wr, &b[k].x, &b[k].y, &b[k].vx, &b[k].vy);
        {
            const R vx0 = ball->vx, vy0 = ball->vy;
            int hit = SUF(walls)(e, e->r_ball, e->e_wb, &ball->x, &ball->y, &ball->vx, &ball->vy);
            if (hit) SUF(ball_wall_spin)(e, hit, vx0, vy0, &ball->vx, &ball->vy, &ball->om);
        }
    }
    /* ---- store ---- */
    s[0] = ball->x; s[1] = ball->y; s[2] = e->r_ball + ball->z; s[3] = ball->vx; s[4] = ball->vy;
    s[e->state_dim] = ball->vz;
    s[e->state_dim + 1] = ball->om;
    for (int k = 0; k < N; ++k) {
        R* r = s + 5 + RS * k;
        const SUF(body)* o = &b[k];
        r[0] = o->x; r[1] = o->y; r[2] = o->th; r[3] = o->vx; r[4] = o->vy;
        r[5] = o->om * e->rad2deg;
        if (ssl) {
            r[6] = c->n_sub ? (o->ir ? RC(1) : RC(0)) : r[6];
            R vf = o->vx * o->c + o->vy * o->s;
            R vl = o->vy * o->c - o->vx * o->s;
            for (int i = 0; i < 4; ++i)
                r[7 + i] = ((vl * e->wc[i] - vf * e->ws[i]) + o->om * e->r_robot) * e->inv_rw;
        }
    }
}

void SUF(rsxo_step)(void* p, const double* cmds) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    R q[MAXROB * 8];
    int n = e->cfg.n_robots * (e->cfg.kind == 0 ? 2 : 8);
    for (int i = 0; i < n; ++i) q[i] = RC(cmds[i]);
    SUF(step_core)(e, q);
}

/* 24-bit uniform in [0, 1) */
static inline R SUF(u01)(uint32_t x) { return RC(x >> 8) * RC(5.9604644775390625e-08); }

/* robosim.step() with commands drawn on the spot (the device-side random-command mode of the raw
 * simulator, rsx_step_dev_random): robot k takes block (env_id, tick, k, RAW) of Philox keyed by seed.
 * VSS: wheel speeds U(-1, 1) * w_max; SSL: local velocities U(-1, 1) * (2.5 m/s, 2.5 m/s, 10 rad/s)
 *  */
void SUF(rsxo_step_random)(void* p, uint64_t seed, uint64_t env_id, uint32_t tick) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    const int N = e->cfg.n_robots, ssl = e->cfg.kind == 1;
    R q[MAXROB * 8];
    memset(q, 0, sizeof(q));
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int k = 0; k < N; ++k) {
        uint32_t ctr[4] = {(uint32_t)env_id, tick, (uint32_t)k, RSXO_DOM_RAW}, u[4];
        rsxo_philox4x32_7(ctr, key, u);
        R a0 = SUF(u01)(u[0]) * RC(2) - RC(1), a1 = SUF(u01)(u[1]) * RC(2) - RC(1), a2 = SUF(u01)(u[2]) * RC(2) - RC(1);
        if (ssl) { q[8 * k + 1] = a0 * RC(2.5); q[8 * k + 2] = a1 * RC(2.5); q[8 * k + 3] = a2 * RC(10.0); }
        else { q[2 * k] = a0 * e->w_max; q[2 * k + 1] = a1 * e->w_max; }
    }
    SUF(step_core)(e, q);
}

/* ==========================================================================================
 * TASKS
 * ======================================================================================== */

/* per-step draws (actions, OU noise): keyed by the number of step() calls since attach, NOT by the env's
 * episode / step counters — the engine can then draw while those counters are still on their way from
 * memory (they are only needed for TimeLimit and placement) */
static void SUF(draw_step)(const SUF(rsxo_env)* e, uint32_t dom, uint32_t out[4]) {
    uint32_t ctr[4] = {e->env_id, 0u, e->tick, dom};
    rsxo_philox4x32_7(ctr, e->key, out);
}
/* placement draws: keyed by (episode, index of the draw) */
static void SUF(draw)(const SUF(rsxo_env)* e, uint32_t tick, uint32_t dom, uint32_t out[4]) {
    uint32_t ctr[4] = {e->env_id, e->episode, tick, dom};
    rsxo_philox4x32_7(ctr, e->key, out);
}

int SUF(rsxo_task_attach)(void* p, int task, uint64_t seed, uint64_t env_id, int max_steps) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    const rsxo_cfg* c = &e->cfg;
    const double* f = c->field;
    if (task == 1) {
        if (c->kind != 0 || c->n_blue < 1) return -1;
        e->obs_dim = 4 + 7 * c->n_blue + 5 * c->n_yellow; e->act_dim = 2; e->info_dim = 6;
        e->max_steps = max_steps > 0 ? max_steps : 1200;
    } else if (task == 2) {
        if (c->kind != 1 || c->n_blue != 1) return -1;
        e->obs_dim = 4 + 8 * c->n_blue + 2 * c->n_yellow; e->act_dim = 5; e->info_dim = 8;
        e->max_steps = max_steps > 0 ? max_steps : 1000;
    } else if (task == 3) {  /* SSLDribbling-v0: dribbling.py:46-57, registry 4800 steps */
        if (c->kind != 1 || c->n_blue != 1 || c->n_yellow != 4) return -1;
        e->obs_dim = 5 + 8 * c->n_blue + 2 * c->n_yellow; e->act_dim = 4; e->info_dim = 1;
        e->max_steps = max_steps > 0 ? max_steps : 4800;
    } else if (task == 4) {  /* SSLContestedPossession-v0: contested_possession.py:42-52, 1200 steps */
        if (c->kind != 1 || c->n_blue != 1 || c->n_yellow != 1) return -1;
        e->obs_dim = 4 + 8 * c->n_blue + 2 * c->n_yellow; e->act_dim = 5; e->info_dim = 9;
        e->max_steps = max_steps > 0 ? max_steps : 1200;
    } else if (task == 5) {  /* SSLPassEndurance-v0: pass_endurance.py:45-56, 1200 steps */
        if (c->kind != 1 || c->n_blue != 2 || c->n_yellow != 0) return -1;
        e->obs_dim = 4 + 6 * c->n_blue; e->act_dim = 3; e->info_dim = 2;
        e->max_steps = max_steps > 0 ? max_steps : 1200;
    } else if (task == 6 || task == 7) {  /* synthetic scrimmage (every robot commanded), spread / crowded line-up */
        if (c->kind != 1 || c->n_robots < 1) return -1;
        e->obs_dim = 2 + 2 * c->n_robots; e->act_dim = 4 * c->n_robots; e->info_dim = 2;
        e->max_steps = max_steps > 0 ? max_steps : 1200;
    } else return -1;
    e->task = task;
    e->key[0] = (uint32_t)seed; e->key[1] = (uint32_t)(seed >> 32);
    e->env_id = (uint32_t)env_id; e->episode = 0xFFFFFFFFu; e->steps = 0; e->tick = 0;
    /* normalisers — vss_gym_base.py:52-58 / ssl_gym_base.py:53-59 */
    double max_pos = fmax(f[1] / 2, f[0] / 2 + f[2]);
    double max_v = (f[16] / 60.0) * 2.0 * RSXO_PI * f[15];
    double max_w = (max_v / (c->kind == 0 ? 0.04 : 0.095)) * (180.0 / RSXO_PI);
    if (task >= 2) { max_v = 2.5; max_w = 10.0; } /* static_defenders.py:76-77 and the same in the other SSL tasks */
    e->max_pos = RC(max_pos); e->inv_max_pos = RC(1.0 / max_pos);
    e->max_v = RC(max_v); e->inv_max_v = RC(1.0 / max_v); e->inv_max_w = RC(1.0 / max_w);
    e->deadzone = RC(0.05);
    double dt = c->time_step_ms * 0.001;
    e->hl_goal = RC(f[0] / 2.0 + f[5]); e->inv_len_cm = RC(1.0 / (f[0] * 100.0));
    e->inv_dt = RC(dt > 0 ? 1.0 / dt : 0.0);
    e->pen_x = RC(f[0] / 2 - f[2]); e->half_pen_wid = RC(f[3] / 2);
    e->inv_bd_scale = RC(1.0 / sqrt(f[1] * f[1] + (f[0] / 2) * (f[0] / 2)));
    e->inv_bg_scale = RC(1.0 / (sqrt((f[1] / 2) * (f[1] / 2) + (f[0] / 2) * (f[0] / 2)) / 4.0));
    e->inv_en_scale = RC(1.0 / (160.0 * 4.0 * (task == 4 ? 1200.0 : 1000.0)));  /* static_defenders.py:71-73, contested_possession.py:60-62 */
    if (task == 1) { /* vss_gym.py:199-206 */
        e->pl_xlo = RC(-(f[0] / 2) + 0.1); e->pl_xspan = RC((f[0] / 2 - 0.1) - (-(f[0] / 2) + 0.1));
        e->pl_min_d2 = RC(0.1 * 0.1);
    } else {         /* static_defenders.py:221-225 */
        e->pl_xlo = RC(0.2); e->pl_xspan = RC((f[0] / 2 - 0.1) - 0.2);
        e->pl_min_d2 = RC(0.2 * 0.2);
    }
    e->pl_ylo = RC(-(f[1] / 2) + 0.1); e->pl_yspan = RC((f[1] / 2 - 0.1) - (-(f[1] / 2) + 0.1));
    if (task == 6) { /* jittered grid over the field: >= 0.2 m apart by construction */
        double sx = f[0] / 8.0, sy = f[1] / 5.0, j = 0.3 * (sx < sy ? sx : sy);
        e->sc_sx = RC(sx); e->sc_sy = RC(sy); e->sc_j = RC(j); e->sc_jb = RC(0.1);
    } else if (task == 7) { /* the same grid packed around the ball: worst-case all-pairs contacts */
        e->sc_sx = RC(0.25); e->sc_sy = RC(0.25); e->sc_j = RC(0.02); e->sc_jb = RC(0.02);
    }
    e->ou_theta_dt = RC(0.17 * dt);          /* Utils.py:6,17 */
    e->ou_sig_sqdt = RC(0.5 * sqrt(dt));     /* Utils.py:8,18 */
    memset(e->metrics, 0, sizeof(e->metrics));
    return 0;
}

/* ---- observations: vss_gym.py:93-117 / static_defenders.py:90-112 ---- */
static void SUF(task_obs)(const SUF(rsxo_env)* e, R* o) {
    const rsxo_cfg* c = &e->cfg;
    const R* s = e->state;
    const R lo = RC(-1.2), hi = RC(1.2);
    int n = 0;
    if (e->task >= 6) { /* scrimmage, README.md:88-90 style: positions only */
        o[n++] = SUF(clampr)(s[0] * e->inv_max_pos, lo, hi);
        o[n++] = SUF(clampr)(s[1] * e->inv_max_pos, lo, hi);
        for (int k = 0; k < c->n_robots; ++k) {
            const R* r = s + 5 + e->RS * k;
            o[n++] = SUF(clampr)(r[0] * e->inv_max_pos, lo, hi);
            o[n++] = SUF(clampr)(r[1] * e->inv_max_pos, lo, hi);
        }
        return;
    }
    if (e->task == 3) o[n++] = ((e->prev_pot / RC(6)) * RC(2)) - RC(1);  /* checkpoint progress, dribbling.py:80 */
    o[n++] = SUF(clampr)(s[0] * e->inv_max_pos, lo, hi);
    o[n++] = SUF(clampr)(s[1] * e->inv_max_pos, lo, hi);
    o[n++] = SUF(clampr)(s[3] * e->inv_max_v, lo, hi);
    o[n++] = SUF(clampr)(s[4] * e->inv_max_v, lo, hi);
    for (int k = 0; k < c->n_blue; ++k) {
        const R* r = s + 5 + e->RS * k;
        R sn, cs;
        R_SINCOS(r[2] * e->deg2rad, &sn, &cs);
        o[n++] = SUF(clampr)(r[0] * e->inv_max_pos, lo, hi);
        o[n++] = SUF(clampr)(r[1] * e->inv_max_pos, lo, hi);
        o[n++] = sn; o[n++] = cs;
        if (e->task != 5) {  /* pass_endurance.py:84-90 leaves the linear velocity out */
            o[n++] = SUF(clampr)(r[3] * e->inv_max_v, lo, hi);
            o[n++] = SUF(clampr)(r[4] * e->inv_max_v, lo, hi);
        }
        o[n++] = SUF(clampr)(r[5] * e->inv_max_w, lo, hi);
        if (e->task == 3) o[n++] = r[6] != RC(0) ? RC(1) : RC(-1);       /* dribbling.py:99 */
        else if (e->task >= 2) o[n++] = r[6] != RC(0) ? RC(1) : RC(0);
    }
    for (int k = c->n_blue; k < c->n_robots; ++k) {
        const R* r = s + 5 + e->RS * k;
        o[n++] = SUF(clampr)(r[0] * e->inv_max_pos, lo, hi);
        o[n++] = SUF(clampr)(r[1] * e->inv_max_pos, lo, hi);
        if (e->task == 1) {
            o[n++] = SUF(clampr)(r[3] * e->inv_max_v, lo, hi);
            o[n++] = SUF(clampr)(r[4] * e->inv_max_v, lo, hi);
            o[n++] = SUF(clampr)(r[5] * e->inv_max_w, lo, hi);
        }
    }
}

/* ---- commands ---- */
/* vss_gym.py:235-254 */
static inline R SUF(vss_wheel)(const SUF(rsxo_env)* e, R a) {
    R v = a * e->max_v;
    v = SUF(clampr)(v, -e->max_v, e->max_v);
    if (-e->deadzone < v && v < e->deadzone) v = RC(0);
    return v * e->inv_rw;
}
/* per-robot actions [N][2] -> cmds [N][2] */
static void SUF(vss_cmds)(const SUF(rsxo_env)* e, const R* act, R* cmds) {
    for (int k = 0; k < e->cfg.n_robots; ++k) {
        cmds[2 * k] = SUF(vss_wheel)(e, act[2 * k]);
        cmds[2 * k + 1] = SUF(vss_wheel)(e, act[2 * k + 1]);
    }
}
/* SSL tasks; theta_deg = pre-step heading of blue 0.
 * task 2/4: static_defenders.py:114-148, contested_possession.py:106-137 (v_x, v_y, v_theta, kick, dribbler)
 * task 3:   dribbling.py:106-135 (v_x, v_y, v_theta, dribbler)
 * task 5:   pass_endurance.py:106-130 (v_theta, kick strength, dribbler; receiver: dribbler on) */
static void SUF(sd_cmds)(const SUF(rsxo_env)* e, const R* a, R theta_deg, R* cmds) {
    memset(cmds, 0, sizeof(R) * 8 * e->cfg.n_robots);
    if (e->task == 5) {
        R k = R_FABS(a[1]) > RC(0.5) ? a[1] : RC(0);
        cmds[3] = a[0] * RC(10.0);
        cmds[5] = k * RC(5.0);
        cmds[7] = a[2] > RC(0) ? RC(1) : RC(0);
        cmds[8 + 7] = RC(1);
        return;
    }
    R sn, cs;
    R_SINCOS(theta_deg * e->deg2rad, &sn, &cs);
    R gx = a[0] * e->max_v, gy = a[1] * e->max_v, vth = a[2] * RC(10.0);
    R lx = gx * cs + gy * sn, ly = gy * cs - gx * sn;
    R nrm = R_SQRT(lx * lx + ly * ly);
    if (!(nrm < e->max_v)) { R sc = e->max_v / nrm; lx = lx * sc; ly = ly * sc; }
    cmds[1] = lx; cmds[2] = ly; cmds[3] = vth;
    if (e->task == 3) {
        cmds[7] = a[3] > RC(0) ? RC(1) : RC(0);
    } else {
        cmds[5] = a[3] > RC(0) ? RC(5.0) : RC(0);
        cmds[7] = a[4] > RC(0) ? RC(1) : RC(0);
    }
}

/* ---- reward / done; `last` = pre-step state (the reference's last_frame) ---- */
static void SUF(task_reward)(SUF(rsxo_env)* e, const R* last, const R* cmds, int first_step) {
    const R* s = e->state;
    R reward = RC(0); int done = 0;
    if (e->task == 1) { /* vss_gym.py:144-192,256-311 */
        R bx = s[0], by = s[1];
        if (bx > e->half_len) { e->info[0] += RC(1); e->info[4] += RC(1); reward = RC(10); done = 1; }
        else if (bx < -e->half_len) { e->info[0] -= RC(1); e->info[5] += RC(1); reward = RC(-10); done = 1; }
        else {
            R dx_d = (e->hl_goal + bx) * RC(100), dx_a = (e->hl_goal - bx) * RC(100), dy = by * RC(100);
            R dy2 = RC(2) * (dy * dy);
            R dist_1 = -R_SQRT(dx_a * dx_a + dy2), dist_2 = R_SQRT(dx_d * dx_d + dy2);
            R pot = ((dist_1 + dist_2) * e->inv_len_cm - RC(1)) * RC(0.5);
            R grad = RC(0);
            if (!first_step) grad = SUF(clampr)((pot - e->prev_pot) * RC(3) * e->inv_dt, RC(-5), RC(5));
            e->prev_pot = pot;
            const R* r0 = s + 5;
            R rbx = bx - r0[0], rby = by - r0[1];
            R nrm = R_SQRT(rbx * rbx + rby * rby);
            /* vss_gym.py:298 divides unguarded; a robot exactly on the ball gets no move term here */
            R mv = nrm > RC(0) ? (rbx / nrm) * r0[3] + (rby / nrm) * r0[4] : RC(0);
            R move = SUF(clampr)(mv * RC(2.5), RC(-5), RC(5));
            R energy = -(R_FABS(cmds[0]) + R_FABS(cmds[1]));
            R t_move = RC(0.2) * move, t_grad = RC(0.8) * grad, t_en = RC(2e-4) * energy;
            reward = (t_move + t_grad) + t_en;
            e->info[1] += t_move; e->info[2] += t_grad; e->info[3] += t_en;
        }
    } else if (e->task >= 6) { /* scrimmage, README.md:96-102 style: a goal ends the episode */
        R bx = s[0], by = s[1];
        if (bx > e->half_len && R_FABS(by) < e->ghw) { reward = RC(1); done = 1; e->info[0] += RC(1); }
        else if (bx < -e->half_len && R_FABS(by) < e->ghw) { reward = RC(-1); done = 1; e->info[1] += RC(1); }
    } else if (e->task == 3) { /* dribbling.py:137-185; prev_pot holds checkpoints_count */
        const R* r0 = s + 5;
        R bx = s[0], by = s[1], lby = last[1], rx = r0[0], ry = r0[1];
        for (int k = 1; k < e->cfg.n_robots; ++k) { /* an obstacle was hit */
            const R* ry_ = s + 5 + 11 * k;
            if (R_FABS(ry_[3]) > RC(0.05) || R_FABS(ry_[4]) > RC(0.05)) done = 1;
        }
        if (rx < RC(-3.0) || rx > RC(1.0) || R_FABS(ry) > RC(1.0)) done = 1; /* node_3 - margin, margin */
        else {
            int n = (int)e->prev_pot, passed = 0;
            int down = lby >= RC(0) && by < RC(0), up = lby < RC(0) && by >= RC(0);
            if (n == 0) passed = bx < RC(-0.5) && bx > RC(-1.0) && down;
            else if (n == 1) passed = bx < RC(-1.0) && bx > RC(-1.5) && up;
            else if (n % 2 == 0) {
                int inside = bx < RC(-1.5) && bx > RC(-2.0);
                passed = inside && down;
                if (inside && !down && up) done = 1; /* reversed the last checkpoint */
            } else passed = bx > RC(-3.0) && bx < RC(-2.0) && up;
            if (passed) {
                reward = RC(1);
                e->prev_pot = RC(n + 1);
                if (n >= 2 && n % 2 == 0 && n + 1 == 7) done = 1; /* course completed */
            }
        }
        e->info[0] = e->prev_pot;
    } else if (e->task == 5) { /* pass_endurance.py:132-154,187-233; prev_pot holds stopped_steps */
        const R* sh = s + 5; const R* rc = s + 5 + 11;
        R bx = s[0], by = s[1], lbx = last[0], lby = last[1];
        R ddx = rc[0] - bx, ddy = rc[1] - by, ldx = rc[0] - lbx, ldy = rc[1] - lby;
        R dist = R_SQRT(ddx * ddx + ddy * ddy), last_dist = R_SQRT(ldx * ldx + ldy * ldy);
        if (rc[6] != RC(0)) { reward = RC(1); done = 1; }
        else {
            R g = e->inv_bg_scale * SUF(clampr)(last_dist - dist, RC(-1), RC(1));
            reward = g; e->info[1] += g;
        }
        /* __wrong_ball: centimetre grid (int truncation), then the stall counter */
        int cbx = (int)(bx * RC(100)), cby = (int)(by * RC(100));
        int csx = (int)(sh[0] * RC(100)), csy = (int)(sh[1] * RC(100));
        int crx = (int)(rc[0] * RC(100)), cry = (int)(rc[1] * RC(100));
        int in_x = (crx < csx ? crx : csx) <= cbx && cbx <= (crx > csx ? crx : csx);
        int in_y = (cry < csy ? cry : csy) <= cby && cby <= (cry > csy ? cry : csy);
        if (R_FABS(last_dist - dist) < RC(0.01)) e->prev_pot = e->prev_pot + RC(1); else e->prev_pot = RC(0);
        if (e->prev_pot > RC(20) || !(in_x && in_y)) { reward = reward - RC(1); done = 1; }
        if (done) {
            R rdx = rc[0] - sh[0], rdy = rc[1] - sh[1];
            R dist_robs = R_SQRT(rdx * rdx + rdy * rdy);
            e->info[0] = dist_robs > RC(0) ? (dist_robs - dist) / dist_robs : RC(0);
        }
    } else { /* static_defenders.py:150-212,256-322; contested_possession.py:139-201 */
        const R* r0 = s + 5;
        R bx = s[0], by = s[1], rx = r0[0], ry = r0[1];
        if (e->task == 4) { /* the opponent was moved: collision (contested_possession.py:166-169) */
            const R* y0 = s + 5 + 11;
            if (R_FABS(y0[3]) > RC(0.1) || R_FABS(y0[4]) > RC(0.1)) { e->info[8] += RC(1); done = 1; }
        }
        if (rx < RC(-0.2) || R_FABS(ry) > e->half_wid) { done = 1; e->info[4] += RC(1); }
        else if (rx > e->pen_x && R_FABS(ry) < e->half_pen_wid) { done = 1; e->info[1] += RC(1); }
        else if (bx < RC(0) || R_FABS(by) > e->half_wid) { done = 1; e->info[2] += RC(1); }
        else if (bx > e->half_len) {
            done = 1;
            if (R_FABS(by) < e->ghw) { reward = RC(5); e->info[0] += RC(1); }
            else e->info[3] += RC(1);
        } else {
            const R* l0 = last + 5;
            R lbx = last[0], lby = last[1];
            R ldx = l0[0] - lbx, ldy = l0[1] - lby;
            R cdx = rx - bx, cdy = ry - by;
            R bd = SUF(clampr)(R_SQRT(ldx * ldx + ldy * ldy) - R_SQRT(cdx * cdx + cdy * cdy), RC(-1), RC(1)) * e->inv_bd_scale;
            R lgx = e->half_len - lbx, cgx = e->half_len - bx;
            R bg = SUF(clampr)(R_SQRT(lgx * lgx + lby * lby) - R_SQRT(cgx * cgx + by * by), RC(-1), RC(1)) * e->inv_bg_scale;
            R en = -(((R_FABS(r0[7]) + R_FABS(r0[8])) + R_FABS(r0[9])) + R_FABS(r0[10])) * e->inv_en_scale;
            e->info[5] += bd; e->info[6] += bg; e->info[7] += en;
            reward = (bd + bg) + en;
        }
    }
    e->reward = reward; e->terminated = (uint8_t)done;
}

/* ---- placement: vss_gym.py:194-233 / static_defenders.py:214-254 with Philox draws ---- */
static void SUF(task_place)(SUF(rsxo_env)* e) {
    const rsxo_cfg* c = &e->cfg;
    R* s = e->state;
    memset(s, 0, sizeof(e->state));
    s[2] = e->r_ball;
    uint32_t n = 0, u[4];
    R px[MAXBOD], py[MAXBOD]; int np = 0;
    int first = 0;
    if (e->task >= 6) { /* scrimmage: robot k in cell (k % 6, k / 6) of a 6 x 4 grid, one Philox block each */
        for (int k = 0; k < c->n_robots; ++k) {
            SUF(draw)(e, (uint32_t)k, RSXO_DOM_PLACE, u);
            R* r = s + 5 + 11 * k;
            r[0] = e->sc_sx * (RC(k % 6) - RC(2.5)) + e->sc_j * (SUF(u01)(u[0]) * RC(2) - RC(1));
            r[1] = e->sc_sy * (RC(k / 6) - RC(1.5)) + e->sc_j * (SUF(u01)(u[1]) * RC(2) - RC(1));
            r[2] = RC(360) * SUF(u01)(u[2]);
        }
        SUF(draw)(e, (uint32_t)c->n_robots, RSXO_DOM_PLACE, u);
        s[0] = e->sc_jb * (SUF(u01)(u[0]) * RC(2) - RC(1));
        s[1] = e->sc_jb * (SUF(u01)(u[1]) * RC(2) - RC(1));
        return;
    }
    if (e->task == 3) { /* dribbling.py:187-202: fixed course */
        s[0] = RC(-0.1); s[1] = RC(0);
        s[5 + 2] = RC(180);
        for (int k = 1; k < 5; ++k) { R* r = s + 5 + 11 * k; r[0] = RC(-0.5) * RC(k); r[2] = RC(180); }
        return;
    }
    if (e->task == 4) { /* contested_possession.py:203-220: opponent holds the ball */
        const double* f = c->field;
        SUF(draw)(e, n++, RSXO_DOM_PLACE, u);
        R ex = RC(f[2]) + RC((f[0] / 2 - f[2]) - f[2]) * SUF(u01)(u[0]);
        R ey = RC(-(f[3] / 2)) + RC(f[3]) * SUF(u01)(u[1]);
        s[0] = ex - RC(0.1); s[1] = ey;
        R* y0 = s + 5 + 11; y0[0] = ex; y0[1] = ey; y0[2] = RC(180);
        return;
    }
    if (e->task == 5) { /* pass_endurance.py:156-185: shooter behind the ball, receiver mirrored */
        SUF(draw)(e, n++, RSXO_DOM_PLACE, u);
        R bx = RC(-1.5) + RC(3.0) * SUF(u01)(u[0]);
        R by = RC(1.5) + RC(-3.0) * SUF(u01)(u[1]);
        R side = by < RC(0) ? RC(-1) : RC(1);
        s[0] = bx; s[1] = by;
        R* sh = s + 5; R* rc = s + 5 + 11;
        sh[0] = bx; sh[1] = by + RC(0.115) * side; sh[2] = side > RC(0) ? RC(270) : RC(90);
        R rx = RC(0);
        for (int t = 0; t < 64; ++t) {
            SUF(draw)(e, n++, RSXO_DOM_PLACE, u);
            rx = RC(-1.5) + RC(3.0) * SUF(u01)(u[0]);
            if (!(R_FABS(rx - bx) < RC(1))) break;
        }
        rc[0] = rx; rc[1] = -by;
        rc[2] = (R_ATAN2(rc[1] - sh[1], rc[0] - sh[0]) + RC(3.14159265358979323846)) * e->rad2deg;
        return;
    }
    if (e->task == 2) { /* blue 0 fixed at the origin */
        for (int t = 0; t < 64; ++t) {
            SUF(draw)(e, n++, RSXO_DOM_PLACE, u);
            s[0] = e->pl_xlo + e->pl_xspan * SUF(u01)(u[0]);
            s[1] = e->pl_ylo + e->pl_yspan * SUF(u01)(u[1]);
            if (!(s[0] > e->pen_x && R_FABS(s[1]) < e->half_pen_wid)) break;
        }
        px[np] = s[0]; py[np] = s[1]; ++np;
        px[np] = RC(0); py[np] = RC(0); ++np;
        first = 1;
    } else {
        SUF(draw)(e, n++, RSXO_DOM_PLACE, u);
        s[0] = e->pl_xlo + e->pl_xspan * SUF(u01)(u[0]);
        s[1] = e->pl_ylo + e->pl_yspan * SUF(u01)(u[1]);
        px[np] = s[0]; py[np] = s[1]; ++np;
    }
    for (int k = first; k < c->n_robots; ++k) {
        R x = RC(0), y = RC(0);
        for (int t = 0; t < 64; ++t) {
            SUF(draw)(e, n++, RSXO_DOM_PLACE, u);
            x = e->pl_xlo + e->pl_xspan * SUF(u01)(u[0]);
            y = e->pl_ylo + e->pl_yspan * SUF(u01)(u[1]);
            int ok = 1;
            for (int q = 0; q < np; ++q) {
                R dx = x - px[q], dy = y - py[q];
                if (dx * dx + dy * dy < e->pl_min_d2) ok = 0;
            }
            if (ok) break;
        }
        SUF(draw)(e, n++, RSXO_DOM_PLACE, u);
        R* r = s + 5 + e->RS * k;
        r[0] = x; r[1] = y; r[2] = RC(360) * SUF(u01)(u[0]);
        px[np] = x; py[np] = y; ++np;
    }
}

static void SUF(task_begin_episode)(SUF(rsxo_env)* e) {
    e->steps = 0;
    if (e->task >= 3) e->prev_pot = RC(0); /* checkpoints_count / stopped_steps */
    memset(e->ou, 0, sizeof(e->ou));
    SUF(task_obs)(e, e->obs);
}

void SUF(rsxo_task_reset)(void* p) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    e->episode += 1; /* every reset() opens a new episode id (first one: 0) */
    SUF(task_place)(e);
    memset(e->info, 0, sizeof(e->info)); e->ep_ret = RC(0); e->prev_pot = RC(0);
    SUF(task_begin_episode)(e);
}
void SUF(rsxo_task_reset_to)(void* p, const double* ball, const double* blue, const double* yellow) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    e->episode += 1;
    SUF(rsxo_reset)(p, ball, blue, yellow);
    memset(e->info, 0, sizeof(e->info)); e->ep_ret = RC(0); e->prev_pot = RC(0);
    SUF(task_begin_episode)(e);
}

/* one fused step; action [act_dim] float or NULL = random */
void SUF(rsxo_task_step)(void* p, const float* action) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    const rsxo_cfg* c = &e->cfg;
    const int N = c->n_robots;
    uint32_t u[4];
    const int first_step = e->steps == 0;
    if (first_step) { memset(e->info, 0, sizeof(e->info)); e->ep_ret = RC(0); }
    R a[8];
    SUF(draw_step)(e, RSXO_DOM_ACT, u);   /* block 0 of the step (VSS-v0: robot 1 owns its words 2, 3) */
    if (e->task >= 6) { /* handled per robot below */ }
    else if (action) for (int i = 0; i < e->act_dim; ++i) a[i] = RC(action[i]);
    else {
        for (int i = 0; i < 4 && i < e->act_dim; ++i) a[i] = SUF(u01)(u[i]) * RC(2) - RC(1);
        if (e->act_dim > 4) { /* fifth component: the low bytes u01 leaves unused in words 0..2 of the same block */
            uint32_t w = (u[0] & 0xFFu) | ((u[1] & 0xFFu) << 8) | ((u[2] & 0xFFu) << 16);
            a[4] = SUF(u01)(w << 8) * RC(2) - RC(1);
        }
    }
    R cmds[MAXROB * 8];
    R last[5 + 11 * MAXROB + RSXO_XROWS];
    memcpy(last, e->state, sizeof(last));
    if (e->task == 1) {
        R act[MAXROB * 2];
        act[0] = a[0]; act[1] = a[1];
        for (int k = 1; k < N; ++k) { /* Utils.py:14-21, Box-Muller on Philox: words of block k >> 1 */
            if (k % 2 == 0) SUF(draw_step)(e, RSXO_DOM_ACT | ((uint32_t)(k >> 1) << 8), u);
            const uint32_t w0 = u[2 * (k & 1)], w1 = u[2 * (k & 1) + 1];
            R u1 = RC((w0 >> 8) + 1u) * RC(5.9604644775390625e-08);
            R ang = (SUF(u01)(w1) - RC(0.5)) * RC(6.283185307179586);
            R rad = R_SQRT(RC(-2) * R_LOG(u1));
            R sn, cs;
            R_SINCOS(ang, &sn, &cs);
            R n0 = rad * cs, n1 = rad * sn;
            e->ou[k][0] = (e->ou[k][0] + e->ou_theta_dt * (RC(0) - e->ou[k][0])) + e->ou_sig_sqdt * n0;
            e->ou[k][1] = (e->ou[k][1] + e->ou_theta_dt * (RC(0) - e->ou[k][1])) + e->ou_sig_sqdt * n1;
            act[2 * k] = e->ou[k][0]; act[2 * k + 1] = e->ou[k][1];
        }
        SUF(vss_cmds)(e, act, cmds);
    } else if (e->task >= 6) { /* scrimmage: every robot gets (v_x, v_y, v_theta, kick), block k of the step */
        memset(cmds, 0, sizeof(R) * 8 * N);
        for (int k = 0; k < N; ++k) {
            R q[4];
            if (action) for (int i = 0; i < 4; ++i) q[i] = RC(action[4 * k + i]);
            else {
                SUF(draw_step)(e, RSXO_DOM_ACT | ((uint32_t)k << 8), u);
                for (int i = 0; i < 4; ++i) q[i] = SUF(u01)(u[i]) * RC(2) - RC(1);
            }
            cmds[8 * k + 1] = q[0] * e->max_v; cmds[8 * k + 2] = q[1] * e->max_v; cmds[8 * k + 3] = q[2] * RC(10.0);
            cmds[8 * k + 5] = q[3] > RC(0.9) ? RC(5.0) : RC(0);
        }
    } else {
        SUF(sd_cmds)(e, a, e->state[5 + 2], cmds);
    }
    memcpy(e->last_cmds, cmds, sizeof(R) * N * (c->kind == 0 ? 2 : 8));
    SUF(step_core)(e, cmds);
    SUF(task_obs)(e, e->obs);
    SUF(task_reward)(e, last, cmds, first_step);
    e->steps += 1;
    e->tick += 1;
    e->ep_ret = e->ep_ret + e->reward;
    e->truncated = (uint8_t)(e->steps >= e->max_steps);
    e->metrics[0] += 1;
    if (e->terminated || e->truncated) {
        memcpy(e->final_obs, e->obs, sizeof(e->obs));
        e->metrics[1] += 1;
        if (e->task == 1) { e->metrics[2] += e->info[4] > RC(0); e->metrics[3] += e->info[5] > RC(0); }
        else if (e->task == 2 || e->task == 4) e->metrics[2] += e->info[0] > RC(0);  /* goal */
        else if (e->task == 3) e->metrics[2] += e->info[0] >= RC(7);                  /* course completed */
        else if (e->task >= 6) { e->metrics[2] += e->info[0] > RC(0); e->metrics[3] += e->info[1] > RC(0); }
        else e->metrics[2] += e->terminated && e->state[5 + 11 + 6] != RC(0);         /* pass received */
        /* VSS-v0: the return is taken from the cumulative reward terms (no running sum is kept) */
        R ret = e->task == 1 ? ((e->info[1] + e->info[2]) + e->info[3]) + RC(10) * e->info[0] : e->ep_ret;
        e->metrics[4] += (int64_t)llrint((double)(ret * RC(1048576.0)));
        e->metrics[5] += e->steps;
        e->metrics[6] += e->truncated && !e->terminated;
        e->episode += 1;
        SUF(task_place)(e);
        SUF(task_begin_episode)(e);
    }
}

/* outputs (any pointer may be NULL) */
void SUF(rsxo_task_out)(void* p, double* obs, double* reward, uint8_t* term, uint8_t* trunc,
                        double* info, double* final_obs, int* steps, int64_t* metrics) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    if (obs) for (int i = 0; i < e->obs_dim; ++i) obs[i] = (double)e->obs[i];
    if (final_obs) for (int i = 0; i < e->obs_dim; ++i) final_obs[i] = (double)e->final_obs[i];
    if (reward) *reward = (double)e->reward;
    if (term) *term = e->terminated;
    if (trunc) *trunc = e->truncated;
    if (info) for (int i = 0; i < e->info_dim; ++i) info[i] = (double)e->info[i];
    if (steps) *steps = e->steps;
    if (metrics) memcpy(metrics, e->metrics, sizeof(e->metrics));
}
void SUF(rsxo_task_last_cmds)(void* p, double* out) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    int n = e->cfg.n_robots * (e->cfg.kind == 0 ? 2 : 8);
    for (int i = 0; i < n; ++i) out[i] = (double)e->last_cmds[i];
}

/* ---- pure task functions exposed for the golden-vector tests ---- */
/* observation of the CURRENT state */
void SUF(rsxo_task_obs_eval)(void* p, double* out) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    R o[64];
    SUF(task_obs)(e, o);
    for (int i = 0; i < e->obs_dim; ++i) out[i] = (double)o[i];
}
/* VSS: per-robot actions [N][2] -> cmds [N][2];  SD: action[5] + heading(deg) -> cmds [N][8] */
void SUF(rsxo_task_cmds_eval)(void* p, const double* act, double theta_deg, double* out) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    R a[MAXROB * 2], q[MAXROB * 8];
    int N = e->cfg.n_robots;
    if (e->task == 1) {
        for (int i = 0; i < 2 * N; ++i) a[i] = RC(act[i]);
        SUF(vss_cmds)(e, a, q);
        for (int i = 0; i < 2 * N; ++i) out[i] = (double)q[i];
    } else {
        for (int i = 0; i < e->act_dim; ++i) a[i] = RC(act[i]);
        SUF(sd_cmds)(e, a, RC(theta_deg), q);
        for (int i = 0; i < 8 * N; ++i) out[i] = (double)q[i];
    }
}
/* reward/done of the transition last -> CURRENT state with sent commands cmds; updates the
 * cumulative info and the ball-potential memory exactly as a step would */
void SUF(rsxo_task_reward_eval)(void* p, const double* last, const double* cmds, int first_step,
                                double* reward, int* done) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    R l[5 + 11 * MAXROB + RSXO_XROWS], q[MAXROB * 8];
    int N = e->cfg.n_robots, C = e->cfg.kind == 0 ? 2 : 8;
    for (int i = 0; i < e->state_dim; ++i) l[i] = RC(last[i]);
    for (int i = 0; i < N * C; ++i) q[i] = RC(cmds[i]);
    if (first_step) { memset(e->info, 0, sizeof(e->info)); }
    SUF(task_reward)(e, l, q, first_step);
    *reward = (double)e->reward; *done = e->terminated;
}
/* OU update with supplied normals (Utils.py:14-21): x <- x + theta*(0-x)*dt + sigma*sqrt(dt)*n */
void SUF(rsxo_ou_eval)(void* p, double* x, const double* nrm, int n) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    for (int i = 0; i < n; ++i) {
        R xi = RC(x[i]);
        xi = (xi + e->ou_theta_dt * (RC(0) - xi)) + e->ou_sig_sqdt * RC(nrm[i]);
        x[i] = (double)xi;
    }
}
/* the per-episode task scalar: checkpoints_count (dribbling) / stopped_steps (pass endurance) */
void SUF(rsxo_task_set_scalar)(void* p, double v) { ((SUF(rsxo_env)*)p)->prev_pot = RC(v); }
double SUF(rsxo_task_get_scalar)(void* p) { return (double)((SUF(rsxo_env)*)p)->prev_pot; }
void SUF(rsxo_task_norms)(void* p, double out[3]) {
    SUF(rsxo_env)* e = (SUF(rsxo_env)*)p;
    out[0] = 1.0 / (double)e->inv_max_pos; out[1] = (double)e->max_v; out[2] = 1.0 / (double)e->inv_max_w;
}
/* elementary functions, for direct comparison with the device versions */
void SUF(rsxo_sincos_eval)(double a, double* s, double* c) { R ss, cc; R_SINCOS(RC(a), &ss, &cc); *s = ss; *c = cc; }
double SUF(rsxo_log_eval)(double x) { return (double)R_LOG(RC(x)); }

/* ---- batch helpers for the CPU baseline (bench.py cpu_baseline leg) ---- */
void SUF(rsxo_vec_task_step)(void** envs, int n_envs, int n_steps) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_envs; ++i)
        for (int t = 0; t < n_steps; ++t) SUF(rsxo_task_step)(envs[i], NULL);
}

#undef RC
#undef MAXROB
#undef MAXBOD
