/* UFCx tabulate_tensor for a(u, v) = inner(grad(u), grad(v)) * dx, P1 on an affine tetrahedron
 * (the form of python/benchmarks/bench_periodic.py:84), written by hand in the style FFCx generates:
 * C99, `restrict`, the UFCx argument list (cpp/assemble_matrix.cpp:438-439).  A[4][4] is accumulated into. */
void tabulate_tensor_laplace_p1_tet(double* restrict A, const double* restrict w, const double* restrict c,
                                    const double* restrict coordinate_dofs, const int* restrict entity_local_index,
                                    const uint8_t* restrict quadrature_permutation, void* custom_data)
{
  (void)w; (void)c; (void)entity_local_index; (void)quadrature_permutation; (void)custom_data;
  const double* x = coordinate_dofs;
  const double J[3][3] = {{x[3] - x[0], x[6] - x[0], x[9] - x[0]},
                          {x[4] - x[1], x[7] - x[1], x[10] - x[1]},
                          {x[5] - x[2], x[8] - x[2], x[11] - x[2]}};
  /* cofactors: K = adj(J)^T / det, grad(lambda_{d+1}) = row d of J^-1 */
  const double C0[3] = {J[1][1] * J[2][2] - J[1][2] * J[2][1], J[0][2] * J[2][1] - J[0][1] * J[2][2], J[0][1] * J[1][2] - J[0][2] * J[1][1]};
  const double C1[3] = {J[1][2] * J[2][0] - J[1][0] * J[2][2], J[0][0] * J[2][2] - J[0][2] * J[2][0], J[0][2] * J[1][0] - J[0][0] * J[1][2]};
  const double C2[3] = {J[1][0] * J[2][1] - J[1][1] * J[2][0], J[0][1] * J[2][0] - J[0][0] * J[2][1], J[0][0] * J[1][1] - J[0][1] * J[1][0]};
  const double det = J[0][0] * C0[0] + J[0][1] * C1[0] + J[0][2] * C2[0];
  double G[4][3];
  for (int k = 0; k < 3; ++k)
  {
    G[1][k] = C0[k] / det;
    G[2][k] = C1[k] / det;
    G[3][k] = C2[k] / det;
    G[0][k] = -(G[1][k] + G[2][k] + G[3][k]);
  }
  const double vol = (det < 0 ? -det : det) / 6.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      A[4 * i + j] += vol * (G[i][0] * G[j][0] + G[i][1] * G[j][1] + G[i][2] * G[j][2]);
}
