/* UFCx tabulate_tensor for the pressure part of the slip-wall term of the Stokes demos,
 *     a01 -= inner(outer(n, n) * dot(-p * Identity(3), n), v) * ds  =  + p (n . v) ds
 * (python/tests/test_rectangular_assembly.py:95-96, python/demos/demo_stokes.py:298), test space P2^3 (30 rows,
 * blocked: row = 3 * i + a), trial space P1 (4 columns), exterior facet entity_local_index[0] of an affine tet.
 * Basix conventions: facet f lies opposite vertex f; P2 dofs = vertices, then edges (2,3)(1,3)(1,2)(0,3)(0,2)(0,1). */
void tabulate_tensor_slip_facet_p2p1_tet(double* restrict A, const double* restrict w, const double* restrict c,
                                         const double* restrict coordinate_dofs, const int* restrict entity_local_index,
                                         const uint8_t* restrict quadrature_permutation, void* custom_data)
{
  (void)w; (void)c; (void)quadrature_permutation; (void)custom_data;
  static const int facet_vertices[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
  static const int edges[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  /* degree-4 rule on the reference triangle (Strang-Fix, 6 points), weights sum to 1/2 */
  static const double qa = 0.445948490915965, wa = 0.223381589678011 / 2.0;
  static const double qb = 0.091576213509771, wb = 0.109951743655322 / 2.0;
  const int f = entity_local_index[0];
  const double* x = coordinate_dofs;
  const double* p0 = x + 3 * facet_vertices[f][0];
  const double* p1 = x + 3 * facet_vertices[f][1];
  const double* p2 = x + 3 * facet_vertices[f][2];
  const double e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const double e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  const double scale = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); /* 2 * facet area */
  /* outward: away from the vertex opposite the facet */
  const double* o = x + 3 * f;
  const double side = n[0] * (p0[0] - o[0]) + n[1] * (p0[1] - o[1]) + n[2] * (p0[2] - o[2]);
  for (int d = 0; d < 3; ++d)
    n[d] = (side > 0 ? n[d] : -n[d]) / scale;
  for (int q = 0; q < 6; ++q)
  {
    const double a = q < 3 ? qa : qb, wq = q < 3 ? wa : wb;
    double mu[3] = {a, a, a};
    mu[q % 3] = 1.0 - 2.0 * a;
    double lam[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < 3; ++k)
      lam[facet_vertices[f][k]] = mu[k];
    double phi[10];
    for (int i = 0; i < 4; ++i)
      phi[i] = lam[i] * (2.0 * lam[i] - 1.0);
    for (int e = 0; e < 6; ++e)
      phi[4 + e] = 4.0 * lam[edges[e][0]] * lam[edges[e][1]];
    for (int i = 0; i < 10; ++i)
      for (int a3 = 0; a3 < 3; ++a3)
        for (int j = 0; j < 4; ++j)
          A[(3 * i + a3) * 4 + j] += wq * scale * lam[j] * phi[i] * n[a3];
  }
}
