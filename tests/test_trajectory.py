"""ASE-format trajectory files: byte layout of the ULM container as documented in ase/io/ulm.py,
the trajectory schema of ase/io/trajectory.py, round trip through the independent reader, append mode, and the
optimizer hooks (`Sella(trajectory=name)`, `IRC(trajectory=name)`).  ASE itself is not installable here: parity with
`ase.io.read` is unpinned (sella_amd/trajectory.py header)."""
import json
import struct

import numpy as np
import pytest

from sella_amd.atoms import Atoms, QuadraticCubicModel
from sella_amd.trajectory import Trajectory, UlmReader


def _model_atoms(n=12, seed=3):
    rng = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.linspace(0.5, 3.0, n)
    lam[0] = -0.8
    A = (Q * lam) @ Q.T
    U = rng.normal(size=(2, n))
    U /= np.linalg.norm(U, axis=1)[:, None]
    atoms = Atoms(['Cu', 'H', 'C', 'O'][:n // 3], 0.05 * rng.normal(size=(n // 3, 3)),
                  cell=np.diag([5.0, 6.0, 7.0]), pbc=[True, True, False])
    atoms.calc = QuadraticCubicModel(lambda x: A @ x, U, c=0.02)
    return atoms


def test_ulm_layout_and_round_trip(tmp_path):
    name = str(tmp_path / 'run.traj')
    atoms = _model_atoms()
    frames = []
    with Trajectory(name, 'w', atoms) as traj:
        traj.write()                                            # no results yet: no calculator section
        for k in range(44):                                     # crosses the 1 -> 42 -> 1764 growth of the offset table
            atoms.positions[0, 0] += 0.01
            e, f = atoms.get_potential_energy(), atoms.get_forces()
            traj.write()
            frames.append((atoms.positions.copy(), e, f))
        assert len(traj) == 45
    raw = open(name, 'rb').read()
    # header, field by field (ase/io/ulm.py)
    assert raw[:8] == b'- of Ulm'
    assert raw[8:24] == b'ASE-Trajectory  '
    version, nitems, pos0 = struct.unpack('<3q', raw[24:48])
    assert (version, nitems) == (3, 45)
    assert pos0 % 8 == 0 and pos0 > 48                          # the table was moved when it grew
    offsets = np.frombuffer(raw[pos0:pos0 + 8 * nitems], dtype='<i8')
    assert np.all(np.diff(offsets) > 0)
    # item 0: int64 length + JSON; arrays precede it, 8-byte aligned
    size0 = struct.unpack('<q', raw[offsets[0]:offsets[0] + 8])[0]
    item0 = json.loads(raw[offsets[0] + 8:offsets[0] + 8 + size0])
    assert item0['version'] == 1 and item0['pbc'] == [True, True, False]
    shape, dtype, off = item0['numbers.']['ndarray']
    assert shape == [4] and dtype == 'int64' and off % 8 == 0 and off == 56      # first array right behind the header
    np.testing.assert_array_equal(np.frombuffer(raw[off:off + 32], dtype='<i8'), [29, 1, 6, 8])
    shape, dtype, off = item0['positions.']['ndarray']
    assert shape == [4, 3] and dtype == 'float64' and off % 8 == 0 and off < offsets[0]
    assert np.asarray(item0['cell']).shape == (3, 3)
    assert 'calculator.' not in item0 and 'masses.' in item0
    # later items carry no header keys (ase/io/trajectory.py write_atoms(write_header=False))
    size5 = struct.unpack('<q', raw[offsets[5]:offsets[5] + 8])[0]
    item5 = json.loads(raw[offsets[5] + 8:offsets[5] + 8 + size5])
    assert sorted(item5) == ['calculator.', 'cell', 'positions.']
    assert sorted(item5['calculator.']) == ['energy', 'forces.', 'name']
    # round trip
    with Trajectory(name) as rd:
        assert len(rd) == 45
        first = rd[0]
        assert first.symbols == ['Cu', 'H', 'C', 'O'] and first.calc is None
        np.testing.assert_array_equal(first.pbc, [True, True, False])
        np.testing.assert_array_equal(first.get_masses(), atoms.get_masses())
        for k, (pos, e, f) in enumerate(frames, start=1):
            img = rd[k]
            np.testing.assert_array_equal(img.positions, pos)
            np.testing.assert_array_equal(img.cell, atoms.cell)
            assert img.get_potential_energy() == e
            np.testing.assert_array_equal(img.get_forces(), f)
        assert len(rd[-3:]) == 3 and np.array_equal(rd[-1].positions, frames[-1][0])


def test_append_and_guards(tmp_path):
    name = str(tmp_path / 'a.traj')
    atoms = _model_atoms()
    with Trajectory(name, 'w', atoms) as t:
        t.write()
        t.write()
    with Trajectory(name, 'a', atoms) as t:
        assert len(t) == 2
        atoms.positions += 0.1
        t.write()
        other = _model_atoms()
        other.pbc = np.array([True, True, True])
        with pytest.raises(ValueError):
            t.write(other)
    rd = UlmReader(name)
    assert len(rd) == 3 and 'numbers' in rd.item(0) and 'numbers' not in rd.item(2)
    rd.close()
    with Trajectory(name) as r:
        np.testing.assert_allclose(r[2].positions - r[1].positions, 0.1)
    with open(str(tmp_path / 'junk.traj'), 'wb') as f:
        f.write(b'not a trajectory file at all, just bytes' * 3)
    with pytest.raises(IOError):
        Trajectory(str(tmp_path / 'junk.traj'))


def test_sella_and_irc_write_trajectories(ctx, tmp_path):
    """`Sella(trajectory=name)`: one image per force call (peswrapper.py:409-418), `append_trajectory` continues the
    file (optimize.py:144-150); `IRC(trajectory=name)`: one image per step through the optimizer's observer."""
    from sella_amd import IRC, Sella
    name = str(tmp_path / 'opt.traj')
    atoms = _model_atoms()
    with Sella(atoms, trajectory=name, logfile=None, order=1, eta=1e-5, gamma=0.0, rs='tr', proj_trans=False) as opt:
        opt.run(1e-6, 6)
        ncalls = atoms.calc.ncalls
    with Trajectory(name) as rd:
        assert len(rd) == ncalls
        np.testing.assert_array_equal(rd[-1].positions, atoms.positions)
        assert rd[-1].get_potential_energy() == atoms.get_potential_energy()
    with Sella(atoms, trajectory=name, append_trajectory=True, logfile=None, order=1, eta=1e-5, gamma=0.0, rs='tr',
               proj_trans=False) as opt:
        opt.run(1e-6, 2)
        appended = opt.pes.neval                                # one image per PES.eval(), cached results included
    with Trajectory(name) as rd:
        assert appended > 0 and len(rd) == ncalls + appended
    irc_name = str(tmp_path / 'irc.traj')
    with IRC(atoms, trajectory=irc_name, logfile=None, dx=0.05, eta=1e-5, gamma=0.0) as irc:
        irc.run(fmax=1e-2, steps=3, direction='forward')
        nsteps = irc.nsteps
    with Trajectory(irc_name) as rd:
        assert len(rd) == nsteps + 1
        np.testing.assert_array_equal(rd[-1].positions, atoms.positions)
