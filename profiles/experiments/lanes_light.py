import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'), os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'), os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle')]
import parity
units, _ = parity.stress_units([(40, 400, 'single', 1.6, 1), (47, 300, 'single', 4, 1000), (41, 300, 'par2', 1.6, 1), (42, 300, 'chain3', 1.6, 1), (43, 300, 'par4', 1.6, 1), (44, 260, 'diamond', 2.5, 1), (45, 200, 'chain2', 3, 1000), (46, 150, 'chain5', 1.5, 1)])
r1, r2, _ = parity.check_units(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'hostemu', '_build', 'libtwgpu_emu.so'), units)
print('lanes ok', [int(r['leaves'].sum()) for r in r2])
