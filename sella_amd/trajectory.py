"""ASE-format trajectory files (`.traj`) without ASE — the on-disk format either side of the hot path
(the reference opens `ase.io.trajectory.Trajectory(name, 'w' | 'a', atoms)` at
sella/peswrapper.py:257-261 and sella/optimize/optimize.py:144-150 and calls `.write()` after every energy/force
evaluation, peswrapper.py:409-418).

The container is ASE's ULM format, version 3, little endian (layout as documented in ase/io/ulm.py):

    bytes 0-7    b'- of Ulm'
    bytes 8-23   tag, space padded: 'ASE-Trajectory'
    bytes 24-47  int64 x 3: version (3), nitems, pos0 = file offset of the table of item offsets
    pos0 ...     int64 x capacity: offset of item i; the table starts with room for 1 entry at byte 48 and is
                 re-written at the (8-byte aligned) end of the file with 42 x the room whenever it is full
    item i       int64 length, then that many bytes of JSON.  A key ending in '.' holds either a nested dictionary
                 or {"ndarray": [shape, dtype name, file offset]}: the raw array bytes, 8-byte aligned ('#' padding),
                 written BEFORE the item's JSON

and the trajectory schema on top of it (ase/io/trajectory.py): item 0 carries `version` = 1, `ase_version`, `pbc`,
`numbers` and `masses`, every item carries `positions`, `cell` and — when the
calculator has results for the geometry — a `calculator.` dictionary with `name`, `energy`, `forces`.

PARITY UNPINNED: ASE is not installable in the build container, so a file written here has not been opened by
`ase.io.read` itself; tests/test_trajectory.py checks the byte layout above field by field and round-trips through
the independent reader below.  `Trajectory(name, 'r')` also reads files written by ASE as long as they use the keys
above (constraints, momenta, tags, charges and magnetic moments are skipped).
"""
import json
import os

import numpy as np

MAGIC = b'- of Ulm'
TAG = 'ASE-Trajectory'
ULM_VERSION = 3
GROW = 42
WRITER_ID = 'sella_amd'

_SCALARS = (bool, int, float, str, type(None), list, tuple, dict)


def _i64(*vals):
    return np.array(vals, dtype='<i8').tobytes()


class UlmWriter:
    """Append-only item writer: `write(key=value, ...)` collects one item, `sync()` commits it."""

    def __init__(self, filename, mode='w', tag=TAG):
        if mode not in ('w', 'a'):
            raise ValueError("mode must be 'w' or 'a'")
        fresh = mode == 'w' or not os.path.isfile(filename) or os.path.getsize(filename) == 0
        if fresh:
            self.fd = open(filename, 'wb')
            self.nitems, self.pos0 = 0, 48
            self.offsets = np.full(1, -1, dtype='<i8')
            self.fd.write(MAGIC + '{:16}'.format(tag).encode('ascii') + _i64(ULM_VERSION, 0, 48) + self.offsets.tobytes())
        else:
            self.fd = open(filename, 'r+b')
            head = read_header(self.fd)
            if head['tag'] != tag:
                raise IOError('%s is not a %s file' % (filename, tag))
            self.nitems, self.pos0 = head['nitems'], head['pos0']
            room = 1
            while room < self.nitems:
                room *= GROW
            self.offsets = np.zeros(room, dtype='<i8')
            self.offsets[:self.nitems] = head['offsets']
            self.fd.seek(0, 2)
        self.data = {}

    def _align(self):
        pos = self.fd.tell()
        pad = (-pos) % 8
        if pad:
            self.fd.write(b'#' * pad)
        return pos + pad

    def _put(self, store, name, value):
        if isinstance(value, np.ndarray):
            arr = np.ascontiguousarray(value)
            if arr.dtype.byteorder == '>':
                arr = arr.astype(arr.dtype.newbyteorder('<'))
            off = self._align()
            self.fd.write(arr.tobytes())
            store[name + '.'] = {'ndarray': [list(arr.shape), arr.dtype.name, off]}
        elif isinstance(value, dict) and name.endswith('.'):
            child = {}
            for k, v in value.items():
                self._put(child, k, v)
            store[name] = child
        elif isinstance(value, (np.integer, np.floating, np.bool_)):
            store[name] = value.item()
        elif isinstance(value, _SCALARS):
            store[name] = value
        else:
            raise TypeError('cannot store %r of type %s' % (name, type(value)))

    def write(self, **items):
        for name, value in items.items():
            self._put(self.data, name, value)

    def child(self, key, items):
        """A nested dictionary (`key.` in the JSON)."""
        self._put(self.data, key + '.', items)

    def sync(self):
        start = self.fd.tell()
        blob = json.dumps(self.data, sort_keys=True).encode()
        self.fd.write(_i64(len(blob)) + blob)
        if self.nitems >= len(self.offsets):                      # table full: a larger one at the end of the file
            grown = np.zeros(len(self.offsets) * GROW, dtype='<i8')
            grown[:len(self.offsets)] = self.offsets
            self.pos0 = self._align()
            self.fd.write(grown.tobytes())
            self.fd.seek(40)
            self.fd.write(_i64(self.pos0))
            self.offsets = grown
        self.offsets[self.nitems] = start
        self.fd.seek(self.pos0 + 8 * self.nitems)
        self.fd.write(_i64(start))
        self.nitems += 1
        self.fd.seek(32)
        self.fd.write(_i64(self.nitems))
        self.fd.flush()
        self.fd.seek(0, 2)
        self.data = {}

    def close(self):
        if not self.fd.closed:
            self.fd.close()


def read_header(fd):
    fd.seek(0)
    if fd.read(8) != MAGIC:
        raise IOError('not an ULM file')
    tag = fd.read(16).decode('ascii').rstrip()
    version, nitems, pos0 = (int(v) for v in np.frombuffer(fd.read(24), dtype='<i8'))
    if version > ULM_VERSION or version < 1:
        raise IOError('unsupported ULM version %d' % version)
    fd.seek(pos0)
    offsets = np.frombuffer(fd.read(8 * nitems), dtype='<i8').copy()
    return dict(tag=tag, version=version, nitems=nitems, pos0=pos0, offsets=offsets)


class UlmReader:
    def __init__(self, filename):
        self.fd = open(filename, 'rb')
        self.head = read_header(self.fd)

    def __len__(self):
        return self.head['nitems']

    def _resolve(self, raw):
        out = {}
        for name, value in raw.items():
            if name.endswith('.'):
                if 'ndarray' in value:
                    shape, dtype, off = value['ndarray']
                    dt = np.dtype(dtype).newbyteorder('<')
                    count = int(np.prod(shape)) if len(shape) else 1
                    self.fd.seek(off)
                    value = np.frombuffer(self.fd.read(count * dt.itemsize), dtype=dt).reshape(shape).copy()
                else:
                    value = self._resolve(value)
                name = name[:-1]
            out[name] = value
        return out

    def item(self, i):
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        self.fd.seek(int(self.head['offsets'][i]))
        size = int(np.frombuffer(self.fd.read(8), dtype='<i8')[0])
        return self._resolve(json.loads(self.fd.read(size).decode()))

    def close(self):
        self.fd.close()


class SinglePointResults:
    """Calculator stand-in attached to the images read back: the stored energy and forces, nothing else."""

    def __init__(self, name, energy, forces):
        self.name, self.energy, self.forces = name, energy, forces
        self.ncalls = 0

    def get_potential_energy(self, atoms=None):
        if self.energy is None:
            raise RuntimeError('no energy stored for this image')
        return self.energy

    def get_forces(self, atoms=None):
        if self.forces is None:
            raise RuntimeError('no forces stored for this image')
        return self.forces.copy()


def _calculator_results(atoms):
    """(name, energy, forces) the calculator already holds for the current geometry, or None — never triggers a
    calculation (ASE: `get_property(..., allow_calculation=False)`)."""
    calc = getattr(atoms, 'calc', None)
    if calc is None:
        return None
    name = getattr(calc, 'name', None) or type(calc).__name__.lower()
    if isinstance(calc, SinglePointResults):
        return name, calc.energy, calc.forces
    res = getattr(calc, '_res', None)
    if res is not None and getattr(calc, '_key', None) == atoms.positions.tobytes():
        return name, float(res[0]), -np.asarray(res[1], dtype=float).reshape(-1, 3)
    results = getattr(calc, 'results', None)                     # an ASE calculator, if one is attached
    if isinstance(results, dict) and 'energy' in results and not getattr(calc, 'calculation_required',
                                                                         lambda a, p: False)(atoms, ['energy']):
        forces = results.get('forces')
        return name, float(results['energy']), None if forces is None else np.asarray(forces, dtype=float)
    return None


class TrajectoryWriter:
    """`Trajectory(filename, 'w' | 'a', atoms)`: `.write()` appends the current state of `atoms` as one image."""

    def __init__(self, filename, mode='w', atoms=None, master=None):
        self.atoms = atoms
        self.backend = UlmWriter(filename, mode)
        self._header = None
        if self.backend.nitems > 0:                              # appending: the header of image 0 rules
            rd = UlmReader(filename)
            first = rd.item(0)
            rd.close()
            self._header = (np.asarray(first['pbc'], dtype=bool), np.asarray(first['numbers']))

    def __len__(self):
        return self.backend.nitems

    @property
    def nframes(self):
        return self.backend.nitems

    def write(self, atoms=None):
        at = self.atoms if atoms is None else atoms
        if at is None:
            raise ValueError('no atoms to write')
        b = self.backend
        pbc = np.asarray(at.pbc, dtype=bool)
        numbers = np.asarray(at.get_atomic_numbers(), dtype=np.int64)
        if self._header is None:
            b.write(version=1, ase_version=WRITER_ID, pbc=pbc.tolist(), numbers=numbers)
            # always explicit: the stand-in Atoms class carries its own (short) table of atomic weights
            b.write(masses=np.asarray(at.get_masses(), dtype=float))
            self._header = (pbc.copy(), numbers.copy())
        else:
            if (pbc != self._header[0]).any():
                raise ValueError('Bad periodic boundary conditions!')
            if len(numbers) != len(self._header[1]):
                raise ValueError('Bad number of atoms!')
            if (numbers != self._header[1]).any():
                raise ValueError('Bad atomic numbers!')
        b.write(positions=np.asarray(at.positions, dtype=float), cell=np.asarray(at.cell, dtype=float).tolist())
        res = _calculator_results(at)
        if res is not None:
            name, energy, forces = res
            fields = dict(name=name)
            if energy is not None:
                fields['energy'] = float(energy)
            if forces is not None:
                fields['forces'] = np.asarray(forces, dtype=float)
            b.child('calculator', fields)
        b.sync()

    def close(self):
        self.backend.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class TrajectoryReader:
    """`Trajectory(filename)`: a sequence of images (`len`, indexing, slicing, iteration)."""

    def __init__(self, filename):
        self.backend = UlmReader(filename)
        if self.backend.head['tag'] != TAG:
            raise IOError('This is not a trajectory file!')
        self._first = self.backend.item(0) if len(self.backend) else None

    def __len__(self):
        return len(self.backend)

    def __getitem__(self, i=-1):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        from .atoms import Atoms
        item = self.backend.item(i)
        head = item if 'numbers' in item else self._first
        atoms = Atoms(np.asarray(head['numbers']), item['positions'], cell=np.asarray(item['cell'], dtype=float),
                      pbc=head['pbc'])
        if head.get('masses') is not None:
            atoms.set_masses(head['masses'])
        calc = item.get('calculator')
        if calc is not None:
            atoms.calc = SinglePointResults(calc.get('name'), calc.get('energy'), calc.get('forces'))
        return atoms

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def close(self):
        self.backend.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def Trajectory(filename, mode='r', atoms=None, master=None):
    """ase.io.trajectory.Trajectory's factory: a reader for 'r', a writer for 'w' / 'a'."""
    if mode == 'r':
        return TrajectoryReader(filename)
    return TrajectoryWriter(filename, mode, atoms, master=master)
