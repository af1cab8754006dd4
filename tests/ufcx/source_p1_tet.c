/* UFCx tabulate_tensor for L(v) = c0 * w_h * f * v * dx on P1 tets with f(x) = 1 + 2 x0 - x1 x2 (quadratic) and a P1
 * coefficient w_h (4 packed dofs in w): integrand of degree 4 -> the 11-point degree-4 Keast rule would do; here the
 * 14-point degree-5 rule given as a table, the way FFCx bakes its rules into the generated code. */
void tabulate_tensor_source_p1_tet(double* restrict A, const double* restrict w, const double* restrict c,
                                   const double* restrict coordinate_dofs, const int* restrict entity_local_index,
                                   const uint8_t* restrict quadrature_permutation, void* custom_data)
{
  (void)entity_local_index; (void)quadrature_permutation; (void)custom_data;
  /* Walkington / Keast 14-point rule of degree 5: two S31 orbits and one S22 orbit, weights sum to 1/6 */
  static const double a1 = 0.31088591926330060980, w1 = 0.11268792571801585080 / 6.0;
  static const double a2 = 0.09273525031089122640, w2 = 0.07349304311636194954 / 6.0;
  static const double a3 = 0.04550370412564964949, w3 = 0.04254602077708146644 / 6.0;
  double L[14][4], W[14];
  int n = 0;
  for (int i = 0; i < 4; ++i, ++n)
  {
    for (int k = 0; k < 4; ++k)
      L[n][k] = k == i ? 1.0 - 3.0 * a1 : a1;
    W[n] = w1;
  }
  for (int i = 0; i < 4; ++i, ++n)
  {
    for (int k = 0; k < 4; ++k)
      L[n][k] = k == i ? 1.0 - 3.0 * a2 : a2;
    W[n] = w2;
  }
  for (int i = 0; i < 4; ++i)
    for (int j = i + 1; j < 4; ++j, ++n)
    {
      for (int k = 0; k < 4; ++k)
        L[n][k] = (k == i || k == j) ? a3 : 0.5 - a3;
      W[n] = w3;
    }
  const double* x = coordinate_dofs;
  const double J[3][3] = {{x[3] - x[0], x[6] - x[0], x[9] - x[0]},
                          {x[4] - x[1], x[7] - x[1], x[10] - x[1]},
                          {x[5] - x[2], x[8] - x[2], x[11] - x[2]}};
  double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0])
               + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  if (det < 0)
    det = -det;
  for (int q = 0; q < 14; ++q)
  {
    double p[3] = {0.0, 0.0, 0.0}, wh = 0.0;
    for (int k = 0; k < 4; ++k)
    {
      for (int d = 0; d < 3; ++d)
        p[d] += L[q][k] * x[3 * k + d];
      wh += L[q][k] * w[k];
    }
    const double f = 1.0 + 2.0 * p[0] - p[1] * p[2];
    for (int i = 0; i < 4; ++i)
      A[i] += c[0] * W[q] * det * wh * f * L[q][i];
  }
}
