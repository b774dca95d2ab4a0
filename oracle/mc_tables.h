/* Oracle-side accessors for the packed marching-cubes case table (test infrastructure). */
#ifndef ORA_MC_TABLES_H
#define ORA_MC_TABLES_H
#include "mc_tables_data.h"
/* triTable[cube][q]  (marching_cube_mesher.py:244-499) */
static inline int mc_tri_table(int cube, int q)
{
    int nib = (int)((MC_TRI_PACKED[cube] >> (4 * q)) & 0xFull);
    return nib == 0xF ? -1 : nib;
}
/* edgeTable[cube]  (marching_cube_mesher.py:225-241) == OR of the edges its triangles use */
static inline int mc_edge_table(int cube)
{
    int mask = 0;
    for (int q = 0; q < 15; ++q) { int e = mc_tri_table(cube, q); if (e >= 0) mask |= 1 << e; }
    return mask;
}
#endif
