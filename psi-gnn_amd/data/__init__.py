from .meshdata import MeshData, collate
from .hexmesh import make_hex_problem, make_from_triangulation, hex_n_for_nodes, hex_lattice
