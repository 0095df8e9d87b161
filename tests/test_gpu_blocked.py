"""The temporally blocked kernel (csrc/sb_blocked.hip: K substeps per launch out of LDS and registers, plan in
csrc/sb_blocking.h) against the CPU oracle, bit for bit, for every K, for substep counts that K does not divide, with
plastic yield, breaks and delete passes, with arbitrary rest lengths (material mode 1), on the floor (acceleration
flags of neighbouring tiles), and the fall-back to the single-substep kernel where the blocked one does not apply."""
import numpy as np
import pytest

from test_gpu_parity import OFF, TILED, assert_same

pytestmark = pytest.mark.gpu


def both(sb, oracle, buf, *, K, n=None, frames=0, bounds=1000.0, tile=256, subticks=64, before=None, calls=1):
    eng = sb.Engine(bounds_size=bounds, subticks=subticks, layout=buf.layout, max_particles=buf.max_particles,
                    max_beams=buf.max_beams, collision_mode=OFF, path=TILED, tile_particles=tile, block_substeps=K)
    ref = oracle.OracleEngine(bounds, 10.0, subticks, buf.layout, OFF, threads=8)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    if before:
        before(eng, ref)
    for _ in range(frames):
        eng.frame()
        ref.frame()
    for _ in range(calls if n else 0):
        eng.step(n)
        ref.step(n)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    info = {k: eng.info(k) for k in ("substeps_per_launch", "tiles", "material_mode", "region_particles")}
    eng.destroy()
    return got, exp, info


@pytest.mark.parametrize("K", [2, 3, 4, 5, 8])
def test_every_block_depth_matches_the_oracle(sb, oracle, K):
    """48x40 lattice with the anti-diagonal beams (4 per particle), 20 tiles, thrown at the floor so that the bottom
    tiles carry border accelerations their neighbours must read; 37 + 37 substeps (no K divides 37), with user input
    and a non-integer drag exponent."""
    buf = sb.scenes.lattice_buffers(48, 40, d=25.0, origin=(60.0, 14.0), spring=50.0, damp=700.0, yield_strain=0.2,
                                    strain_limit=1e9, anti_diagonal=True, jitter=2.0, velocity=(3.0, -9.0), layout=2)
    buf.user_strength = 1.5

    def before(eng, ref):
        b = buf.copy()
        b.set_user_input(applied_force=(0.3, 0.1), mouse_pos=(200.0, 150.0), mouse_vel=(4.0, 2.0), mouse_active=True)
        for e in (eng, ref):
            e.write_user_input(b.user_input_bytes())
            e.set_physics_constants(np.array([0.1, -0.8, 0.4, 0.3, 0.6, 0.2, 0.002, 2.5], "f4"))

    got, exp, info = both(sb, oracle, buf, K=K, n=37, calls=2, before=before)
    assert info["substeps_per_launch"] == K and info["tiles"] >= 8
    assert_same(got, exp, "blocked K=%d" % K)
    assert (exp.particles[:, 1] == 10.0).any() and (exp.particles[:, 4:6] != 0).any()   # floor contact left accelerations behind


def test_yield_break_delete_blocked(sb, oracle):
    """Plastic yield, break flags and the per-frame delete pass with entries of a deleted beam living in several tiles."""
    buf = sb.scenes.lattice_buffers(14, 12, d=30.0, origin=(30.0, 30.0), spring=50.0, damp=100.0, yield_strain=0.05,
                                    strain_limit=0.12, layout=1, velocity=(-40.0, -35.0), slack=8)
    for K in (3, 5):
        got, exp, info = both(sb, oracle, buf, K=K, frames=3, n=5, tile=64)
        assert info["substeps_per_launch"] == K
        assert exp.beam_count < buf.beam_count, "scene must break beams"
        assert_same(got, exp, "break, blocked K=%d" % K)
    assert (exp.beams["target_length"][:exp.beam_count] != exp.beams["length"][:exp.beam_count]).any()


def test_arbitrary_rest_lengths_use_material_mode_1(sb, oracle):
    buf = sb.scenes.lattice_buffers(30, 30, d=25.0, origin=(100.0, 100.0), jitter=3.0, layout=2, strain_limit=1e9)
    B = buf.beam_count
    L = (buf.beams["length"][:B] * (1.0 + 0.05 * sb.scenes.hash_uniform(9, B))).astype("f4")   # every beam its own rest length
    buf.beams["length"][:B] = L
    buf.beams["target_length"][:B] = L
    got, exp, info = both(sb, oracle, buf, K=4, n=50)
    assert info["material_mode"] == 1 and info["substeps_per_launch"] == 4
    assert_same(got, exp, "material mode 1, blocked")


def test_negative_yield_takes_the_single_substep_kernel(sb, oracle):
    """sb_beam_group applies sign(strain) as a copied sign bit, which is the WGSL product only for yield_strain >= 0;
    a scene with a negative yield_strain (every beam 'yields' every substep) must be routed to the kernel that spells
    sign() out -- and still match."""
    buf = sb.scenes.lattice_buffers(20, 20, d=25.0, origin=(100.0, 100.0), jitter=2.0, layout=2, yield_strain=-0.01, strain_limit=1e9)
    got, exp, info = both(sb, oracle, buf, K=4, n=20)
    assert info["substeps_per_launch"] == 1
    assert_same(got, exp, "negative yield")


@pytest.mark.parametrize("spring,damp,blocked", [(1.0e-20, 700.0, False), (50.0, 1.0e32, False), (0.0, 700.0, True), (-50.0, 700.0, True),
                                                 (3.0e-15, 1.0e-14, True), (2.0e29, 0.0, True)])
def test_springs_at_the_edges_of_the_exact_range(sb, oracle, spring, damp, blocked):
    """The blocked kernel multiplies the force scale 65536 into spring and damp (one packed multiplication less per beam): exact
    for zero and for magnitudes between 2^-50 and 2^100 -- negative ones included -- and anything else takes the single-substep
    kernel.  Either way the oracle's bits (forces that saturate the fixed-point sums included: spring 2e29)."""
    buf = sb.scenes.lattice_buffers(24, 20, d=25.0, origin=(100.0, 100.0), jitter=2.0, layout=2, spring=spring, damp=damp,
                                    yield_strain=0.3, strain_limit=1e9, velocity=(0.5, -1.0))
    got, exp, info = both(sb, oracle, buf, K=5, n=23)
    assert (info["substeps_per_launch"] > 1) == blocked, info
    assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")) and got.beams.tobytes() == exp.beams.tobytes()


def test_fallback_with_automatic_tile_size_keeps_the_particle_order(sb, oracle):
    """The blocked plan picks its own tile size (here 256-particle tiles for 90 000 particles); when the scene then turns out
    not to fit the blocked kernel (negative yield), the single-substep tiling must be made on the SAME bisection -- the
    particles are already on the device in that order.  (r02: it was made for 1024-particle tiles.)"""
    buf = sb.scenes.lattice_buffers(300, 300, d=25.0, origin=(100.0, 100.0), jitter=2.0, layout=2, yield_strain=-0.01, strain_limit=1e9,
                                    velocity=(1.0, -2.0))
    got, exp, info = both(sb, oracle, buf, K=0, n=12, bounds=9000.0, tile=0)
    assert info["substeps_per_launch"] == 1 and info["tiles"] > 200
    assert_same(got, exp, "fallback, automatic tile size")


def test_block_depth_is_lowered_until_the_region_fits(sb, oracle):
    """K = 8 on 700-particle tiles of a 4-beam lattice needs more halo than the kernel's slots hold (1536 halo particles,
    3072 halo entries, beside 1024 own particles and 3072 own beams): the engine lowers K and says so."""
    buf = sb.scenes.lattice_buffers(96, 96, d=25.0, origin=(100.0, 100.0), anti_diagonal=True, jitter=1.0, layout=2, strain_limit=1e9)
    got, exp, info = both(sb, oracle, buf, K=8, n=24, bounds=4000.0, tile=700)
    assert 1 < info["substeps_per_launch"] < 8 and info["region_particles"] <= 2560
    assert_same(got, exp, "lowered K")
