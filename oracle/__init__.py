"""CPU oracle for the MF-MDM denoiser + DDPM reverse loop.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and
there only as the checker (or as the timed CPU baseline) - the product path
(oakink2-tamf_amd/) never imports, links or executes anything under oracle/ and fails
loudly when the HIP library is missing.

Parity status: PINNED.  The reference holds no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by oracle/capture_golden.py (committed) and stored as
small fixtures under tests/golden/.  tests/test_oracle_golden.py re-checks the oracle
against every fixture on CPU.

Modules: mdm_oracle (denoiser, DDPM loop, Philox), geometry_oracle (pose decode, hand->object
distance, contact distance, point-in-mesh / SIV, rigid point transforms), det / fixtures
(deterministic inputs shared with the capture script), capture_golden (runs the reference).
oracle/build_ref.sh compiles the one compiled piece of the reference these rows touch - the
Cython TriangleHash of the SIV score - from the reference's own source into oracle/_ref/
(git-ignored); it is used by capture_golden.py only.
"""
