"""Prism / tesseroid meshes (host side).

Keeps the class and method surface of the reference's `mesher` package
(mesher/geometry.py:51-210, mesher/mesh.py:126-955) so that driver scripts keep working,
but is organised around what the device path needs: one vectorised table of the active
cells' bounds (`cell_bounds()`, M x 6, mesh order = z slowest, then y, x fastest, carved
cells removed -- the order gravmag/prism.py:299-312 walks the mesh in).  The per-cell
floating-point expressions are the reference's (mesh.py:237-266, :659-681) evaluated
with the same operation order, so bounds are bit-identical.

Differences by design: the carve mask is a boolean array (the reference keeps a Python
list and does `index in self.mask`, O(len(mask)) per access); `mask`
remains available as the list of carved flat indices for callers that read it.
"""
import copy as _cp

import numpy as np


class GeometricElement(object):
    """Base of Prism/Tesseroid (geometry.py:13-43)."""

    def __init__(self, props):
        self.props = {}
        if props is not None:
            for p in props:
                self.props[p] = props[p]

    def addprop(self, prop, value):
        self.props[prop] = value

    def copy(self):
        return _cp.deepcopy(self)


class Prism(GeometricElement):
    """Right rectangular prism, x north / y east / z down (geometry.py:51-106)."""

    def __init__(self, x1, x2, y1, y2, z1, z2, props=None):
        super().__init__(props)
        self.x1, self.x2 = float(x1), float(x2)
        self.y1, self.y2 = float(y1), float(y2)
        self.z1, self.z2 = float(z1), float(z2)

    def __str__(self):
        names = [('x1', self.x1), ('x2', self.x2), ('y1', self.y1),
                 ('y2', self.y2), ('z1', self.z1), ('z2', self.z2)]
        names.extend((p, self.props[p]) for p in sorted(self.props))
        return ' | '.join('%s:%g' % (n, v) for n, v in names)

    def get_bounds(self):
        return [self.x1, self.x2, self.y1, self.y2, self.z1, self.z2]

    def center(self):
        return np.array([0.5 * (self.x1 + self.x2), 0.5 * (self.y1 + self.y2),
                         0.5 * (self.z1 + self.z2)])


class Tesseroid(GeometricElement):
    """Spherical prism: w,e,s,n in degrees, top/bottom heights in m (geometry.py:109-210)."""

    def __init__(self, w, e, s, n, top, bottom, props=None):
        super().__init__(props)
        self.w, self.e = float(w), float(e)
        self.s, self.n = float(s), float(n)
        self.bottom, self.top = float(bottom), float(top)

    def __str__(self):
        names = [('w', self.w), ('e', self.e), ('s', self.s),
                 ('n', self.n), ('top', self.top), ('bottom', self.bottom)]
        names.extend((p, self.props[p]) for p in sorted(self.props))
        return ' | '.join('%s:%g' % (n, v) for n, v in names)

    def get_bounds(self):
        return [self.w, self.e, self.s, self.n, self.top, self.bottom]

    def half(self, lon=True, lat=True, r=True):
        dlon = 0.5 * (self.e - self.w)
        dlat = 0.5 * (self.n - self.s)
        dh = 0.5 * (self.top - self.bottom)
        wests = [self.w, self.w + dlon]
        souths = [self.s, self.s + dlat]
        bottoms = [self.bottom, self.bottom + dh]
        if not lon:
            dlon *= 2
            wests.pop()
        if not lat:
            dlat *= 2
            souths.pop()
        if not r:
            dh *= 2
            bottoms.pop()
        return [Tesseroid(i, i + dlon, j, j + dlat, k + dh, k, props=self.props)
                for i in wests for j in souths for k in bottoms]

    def split(self, nlon, nlat, nh):
        wests = np.linspace(self.w, self.e, nlon + 1)
        souths = np.linspace(self.s, self.n, nlat + 1)
        bottoms = np.linspace(self.bottom, self.top, nh + 1)
        dlon, dlat, dh = wests[1] - wests[0], souths[1] - souths[0], bottoms[1] - bottoms[0]
        return [Tesseroid(i, i + dlon, j, j + dlat, k + dh, k, props=self.props)
                for i in wests[:-1] for j in souths[:-1] for k in bottoms[:-1]]


class _MeshBase(object):
    """Shared machinery: iteration, masking, vectorised bounds, topography carving."""

    celltype = Prism
    _topo_method = 'cubic'

    def _finish_init(self, shape, props):
        self.shape = tuple(int(i) for i in shape)
        self.size = int(self.shape[0] * self.shape[1] * self.shape[2])
        self.props = {} if props is None else props
        self.i = 0
        self.mask = []                       # carved flat indices (reference attribute)
        self._carved = np.zeros(self.size, dtype=bool)
        self.zdown = True

    # -- layer geometry: (z1, z2) of layer k, reference arithmetic -------------------
    def _layer_z(self, k):
        raise NotImplementedError

    def _layer_table(self):
        nz = self.shape[0]
        z = np.empty((nz, 2))
        for k in range(nz):
            z[k] = self._layer_z(k)
        return z

    # -- list-like protocol -----------------------------------------------------------
    def __len__(self):
        return self.size

    def __iter__(self):
        self.i = 0
        return self

    def __next__(self):
        if self.i >= self.size:
            raise StopIteration
        cell = self.__getitem__(self.i)
        self.i += 1
        return cell

    def __getitem__(self, index):
        if index >= self.size or index < -self.size:
            raise IndexError('mesh index out of range')
        if index < 0:
            index = self.size + index
        if self._carved[index]:
            return None
        nz, ny, nx = self.shape
        k = index // (nx * ny)
        j = (index - k * (nx * ny)) // nx
        i = (index - k * (nx * ny) - j * nx)
        x1 = self.bounds[0] + self.dims[0] * i
        x2 = x1 + self.dims[0]
        y1 = self.bounds[2] + self.dims[1] * j
        y2 = y1 + self.dims[1]
        z1, z2 = self._layer_z(k)
        props = dict([p, self.props[p][index]] for p in self.props)
        return self.celltype(x1, x2, y1, y2, z1, z2, props=props)

    def addprop(self, prop, values):
        self.props[prop] = values

    def copy(self):
        return _cp.deepcopy(self)

    # -- what the device path consumes ------------------------------------------------
    def active_index(self):
        """Flat indices (mesh order) of the cells that are not carved."""
        return np.flatnonzero(~self._carved)

    def cell_bounds(self, active_only=True):
        """(M, 6) float64 table x1,x2,y1,y2,z1,z2 (or w,e,s,n,top,bottom) in mesh order."""
        nz, ny, nx = self.shape
        x1 = self.bounds[0] + self.dims[0] * np.arange(nx, dtype=np.float64)
        x2 = x1 + self.dims[0]
        y1 = self.bounds[2] + self.dims[1] * np.arange(ny, dtype=np.float64)
        y2 = y1 + self.dims[1]
        z = self._layer_table()
        out = np.empty((nz, ny, nx, 6))
        out[..., 0] = x1[None, None, :]
        out[..., 1] = x2[None, None, :]
        out[..., 2] = y1[None, :, None]
        out[..., 3] = y2[None, :, None]
        out[..., 4] = z[:, 0][:, None, None]
        out[..., 5] = z[:, 1][:, None, None]
        out = out.reshape(self.size, 6)
        if active_only and self._carved.any():
            out = out[~self._carved]
        return np.ascontiguousarray(out)

    # -- topography -------------------------------------------------------------------
    def _carve_levels(self):
        raise NotImplementedError

    def carvetopo(self, x, y, height, below=False, save_interp=None):
        """Mask cells above (or below) a topography surface (mesh.py:301-394, :729-801).

        Same decision rule as the reference (level `cellz` of every layer against the
        interpolated surface), evaluated as one array comparison.  `save_interp`: optional
        file name for the interpolated surface (the reference always writes
        'carve_topo_interp.txt' into the working directory)."""
        import scipy.interpolate
        nz, ny, nx = self.shape
        x1, x2, y1, y2 = self.bounds[:4]
        dx, dy = self.dims[0], self.dims[1]
        xc = np.arange(x1, x2, dx) + 0.5 * dx
        if len(xc) > nx:
            xc = xc[:-1]
        yc = np.arange(y1, y2, dy) + 0.5 * dy
        if len(yc) > ny:
            yc = yc[:-1]
        zc = np.asarray(self._carve_levels(), dtype=np.float64)
        if len(zc) > nz:
            zc = zc[:-1]
        XC, YC = np.meshgrid(xc, yc)
        topo = scipy.interpolate.griddata((x, y), height, (XC, YC),
                                          method=self._topo_method).ravel()
        if self.zdown:
            topo = -1 * topo
        if save_interp:
            np.savetxt(save_interp, np.c_[XC.ravel(), YC.ravel(), topo], fmt='%.8f',
                       delimiter=' ')
        if np.ma.isMA(topo):
            tmask = np.ma.getmaskarray(topo)
            topo = np.ma.getdata(topo)
        else:
            tmask = np.zeros(topo.shape, dtype=bool)
        cz = zc[:, None]
        h = topo[None, :]
        with np.errstate(invalid='ignore'):
            if below:
                hit = (cz > h) if self.zdown else (cz < h)
            else:
                hit = (cz < h) if self.zdown else (cz > h)
        hit = hit | tmask[None, :]
        flat = np.flatnonzero(hit.ravel())
        self.mask.extend(int(c) for c in flat)
        self._carved[flat] = True
        return self.mask

    # -- node coordinates -------------------------------------------------------------
    def get_xs(self):
        x1, x2 = self.bounds[0], self.bounds[1]
        dx = self.dims[0]
        xs = np.arange(x1, x2 + dx, dx)
        if xs.size > self.shape[2] + 1:
            return xs[:-1]
        return xs

    def get_ys(self):
        y1, y2 = self.bounds[2], self.bounds[3]
        dy = self.dims[1]
        ys = np.arange(y1, y2 + dy, dy)
        if ys.size > self.shape[1] + 1:
            return ys[:-1]
        return ys

    def get_layer(self, i):
        nz, ny, nx = self.shape
        if i >= nz or i < 0:
            raise IndexError('Layer index %d is out of range.' % (i))
        return [self.__getitem__(p) for p in range(i * nx * ny, (i + 1) * nx * ny)]

    def layers(self):
        for i in range(self.shape[0]):
            yield self.get_layer(i)


class PrismMesh(_MeshBase):
    """Regular mesh of prisms, optional geometric growth of dz (mesh.py:126-516).

    bounds = [xmin, xmax, ymin, ymax, zmin, zmax]; spacing = (dz, dy, dx); ratio >= 1."""

    celltype = Prism
    _topo_method = 'cubic'

    def __init__(self, bounds, spacing, ratio=1, props=None, verbose=False):
        dz, dy, dx = spacing
        x1, x2, y1, y2, z1, z2 = bounds
        self.dims = (dx, dy, dz)
        self.ratio = ratio
        nx = int(np.ceil((x2 - x1) / dx))
        ny = int(np.ceil((y2 - y1) / dy))
        if ratio == 1:
            nz = int(np.ceil((z2 - z1) / dz))
            bounds_big = x1, x1 + nx * dx, y1, y1 + ny * dy, z1, z1 + nz * dz
        else:
            # geometric layer thickness dz*ratio**k: keep adding layers while the running
            # bottom is above zmax and more than dz remains (mesh.py:181-193)
            n = 1
            while True:
                depth = z1 + dz * (1 - ratio ** n) / (1 - ratio)
                if depth < z2 and (z2 - depth) > dz:
                    n += 1
                else:
                    break
            nz = int(n)
            bounds_big = x1, x1 + nx * dx, y1, y1 + ny * dy, z1, z2
        if verbose:
            print("grid boundaries: {}".format(bounds_big))
        self.bounds = bounds_big
        self._finish_init((nz, ny, nx), props)

    def _layer_z(self, k):
        nz = self.shape[0]
        dz = self.dims[2]
        if self.ratio == 1:
            z1 = self.bounds[4] + dz * k
            z2 = z1 + dz if k < nz - 1 else self.bounds[5]
        else:
            z2 = self.bounds[4] + dz * (1 - self.ratio ** (k + 1)) / (1 - self.ratio)
            z1 = z2 - dz * self.ratio ** k
            if k == nz - 1:
                z2 = self.bounds[5]
        return z1, z2

    def _carve_levels(self):
        nz = self.shape[0]
        z1, z2 = self.bounds[4], self.bounds[5]
        dz = self.dims[2]
        if self.ratio == 1:
            return np.arange(z1, z2, dz) + 0.5 * dz
        zc = np.zeros(nz)
        bottom = z1
        for k in range(0, nz - 1):
            bottom = self.bounds[4] + dz * (1 - self.ratio ** (k + 1)) / (1 - self.ratio)
            zc[k] = bottom - 0.5 * dz * self.ratio ** k
        zc[nz - 1] = bottom + 0.5 * (z2 - bottom)
        return zc

    def get_zs(self):
        nz = self.shape[0]
        z1, z2 = self.bounds[4], self.bounds[5]
        dz = self.dims[2]
        if self.ratio == 1:
            zs = np.arange(z1, z2 + dz, dz)
        else:
            zs = np.zeros(nz + 1)
            for k in range(0, nz):
                bottom = self.bounds[4] + dz * (1 - self.ratio ** (k + 1)) / (1 - self.ratio)
                zs[k] = bottom - dz * self.ratio ** k
            zs[nz] = z2
        if zs.size > nz + 1:
            return zs[:-1]
        return zs


class TesseroidMesh(PrismMesh):
    """Mesh of tesseroids: bounds = [w, e, s, n, top, bottom], spacing = (dr, dlat, dlon),
    dr negative (heights decrease with layer index) (mesh.py:518-559)."""

    celltype = Tesseroid

    def __init__(self, bounds, spacing, ratio=1, props=None, verbose=False):
        super().__init__(bounds, spacing, ratio, props, verbose)
        self.zdown = False
        self.dump = None


class PrismMeshSegment(_MeshBase):
    """Mesh whose layer thickness is piecewise constant (mesh.py:561-912).

    spacing = ([dz1, dz2, ...], dy, dx); divisionsection = [z0, z1, ..., zmax] gives the
    depth interval each dz applies to."""

    celltype = Prism
    _topo_method = 'nearest'

    def __init__(self, bounds, spacing, divisionsection, props=None, verbose=False):
        x1, x2, y1, y2, z1, z2 = bounds
        dzlist, dy, dx = spacing
        self.dims = (dx, dy, dzlist)
        self.segment = len(dzlist)
        self.divisionsection = divisionsection
        nx = int(np.ceil((x2 - x1) / dx))
        ny = int(np.ceil((y2 - y1) / dy))
        nz = 0
        nzlist = np.zeros(self.segment)
        nzsumlist = np.zeros(self.segment)
        for i in range(self.segment):
            nzlist[i] = int(np.ceil((divisionsection[i + 1] - divisionsection[i]) / dzlist[i]))
            nz = nz + nzlist[i]
            nzsumlist[i] = nz
        bounds_big = (x1, x1 + nx * dx, y1, y1 + ny * dy, z1,
                      self.divisionsection[-2] + nzlist[-1] * dzlist[-1])
        if verbose:
            print("segment grid boundaries: {}".format(bounds_big))
        self.nzlist = nzlist
        self.nzsumlist = nzsumlist
        self.bounds = bounds_big
        self._finish_init((nz, ny, nx), props)

    def _layer_z(self, k):
        kloc = 0
        for iseg in range(self.segment):
            if k < self.nzsumlist[iseg]:
                kloc = iseg
                break
        dz = self.dims[2][kloc]
        if kloc == 0:
            z1 = self.bounds[4] + dz * k
        else:
            z1 = self.divisionsection[kloc] + dz * (k - self.nzsumlist[kloc - 1])
        return z1, z1 + dz

    def _carve_levels(self):
        zc = []
        dzlist = self.dims[2]
        for iseg in range(self.segment):
            zc.extend(list(np.arange(self.divisionsection[iseg],
                                     self.divisionsection[iseg + 1], dzlist[iseg])))
        return np.array(zc)

    def get_zs(self):
        nz = self.shape[0]
        dzlist = self.dims[2]
        zs = []
        for iseg in range(self.segment):
            zs.extend(list(np.arange(self.divisionsection[iseg],
                                     self.divisionsection[iseg + 1], dzlist[iseg])))
        zs.append(self.bounds[5])
        zs = np.array(zs)
        if zs.size > nz + 1:
            return zs[:-1]
        return zs


class TesseroidMeshSegment(PrismMeshSegment):
    """Piecewise-dr tesseroid mesh (mesh.py:914-955)."""

    celltype = Tesseroid

    def __init__(self, bounds, spacing, divisionsection, props=None, verbose=False):
        super().__init__(bounds, spacing, divisionsection, props, verbose)
        self.zdown = False
        self.dump = None
