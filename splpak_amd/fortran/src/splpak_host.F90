!> Host solver of the drop-in `splpak_module`: the same least-squares problem as the reference's
!! splcw (rows of src/splpak.F90:788-855, sparse-area histogram :862-910, derivative-constraint rows
!! :921-1046) solved on the CPU as banded normal equations + Cholesky + iterative refinement against the
!! rows, written from scratch for this package (it shares nothing with the test oracle under `oracle/`).
!!
!! It is NOT a fallback of the GPU path: `initialize` runs on the MI355X and fails loudly without one.
!! This solver runs only (a) when the caller asks for it explicitly (`call solver%set_host(.true.)`:
!! build machines, CI, small problems next to a busy GPU) and (b) in a -DREAL128 build, which has no GPU
!! arithmetic at all (SURVEY section 8f-4: fpm dependents and quad-precision users keep working).
!! Arithmetic is real64 (real128 in a REAL128 build) whatever the storage kind.
!!
!! Cost: O(ndata * 16^ndim) for the rows, O(ncol * p^2) for the factorisation with the half bandwidth
!! p = 3 * sum_d prod_{e<d} nodes(e) and 8 (p+1) ncol bytes -- fine up to a few 10^4 columns
!! (16^3: 27 MB, 1 s; 32^3: 0.8 GB, minutes), not a replacement of the GPU path for BASELINE's grids.
module splpak_host_solver

    use iso_fortran_env, only: real32, real64, real128

    implicit none

    private

#ifdef REAL32
    integer,parameter :: wp = real32
#elif REAL128
    integer,parameter :: wp = real128
#else
    integer,parameter :: wp = real64
#endif
#ifdef REAL128
    integer,parameter,public :: hk = real128     !! arithmetic kind of the host computations
#else
    integer,parameter,public :: hk = real64
#endif

    public :: host_basis, host_fit

    contains

    !> One 1-D factor of the tensor-product basis: the natural-spline basis function centred on node
    !! `ib` of a dimension with `nod` nodes, spacing 1/s, or its first / second derivative, at `xx`.
    !! Closed forms of SURVEY appendix A (reference bascmp :231-381): interior functions are the
    !! cubic "chapeau" B-splines, the two functions at either end are cubic inside and straight lines
    !! outside (natural boundary, linear extrapolation).  Strict inequalities as in the reference.
    pure function host_basis(ib,nod,ider,xx,xnode,s) result(b)
        integer,intent(in) :: ib, nod, ider
        real(hk),intent(in) :: xx, xnode, s
        real(hk) :: b, z, z1, f
        b = 0.0_hk
        if (ib >= 2 .and. ib <= nod-3) then                 ! chapeau (:253-300)
            select case (ider)
            case (0)
                z = abs(s*(xx-xnode)) - 2.0_hk
                if (z < 0.0_hk) then
                    b = -0.25_hk*z**3
                    z1 = z + 1.0_hk
                    if (z1 < 0.0_hk) b = b + z1**3
                end if
            case (1)
                f = s
                if (xx-xnode < 0.0_hk) f = -s
                z = f*(xx-xnode) - 2.0_hk
                if (z < 0.0_hk) then
                    b = -0.75_hk*z**2
                    z1 = z + 1.0_hk
                    if (z1 < 0.0_hk) b = b + 3.0_hk*z1**2
                    b = b*f
                end if
            case default
                z = s*abs(xx-xnode) - 2.0_hk
                if (z < 0.0_hk) then
                    b = -1.5_hk*z
                    z1 = z + 1.0_hk
                    if (z1 < 0.0_hk) b = b + 6.0_hk*z1
                    b = b*s*s
                end if
            end select
            return
        end if
        f = s                                               ! end functions: left (ib <= 1) mirrors right (:302-379)
        if (ib <= 1) f = -s
        z = f*(xx-xnode) + 2.0_hk
        select case (ider)
        case (0)
            if (z > 0.0_hk) then
                if (z < 2.0_hk) then
                    b = 0.5_hk*z**3
                    z1 = z - 1.0_hk
                    if (z1 > 0.0_hk) b = b - z1**3
                else
                    b = 3.0_hk*z - 3.0_hk
                end if
            end if
        case (1)
            if (z > 0.0_hk) then
                if (z < 2.0_hk) then
                    b = 1.5_hk*z**2
                    z1 = z - 1.0_hk
                    if (z1 > 0.0_hk) b = b - 3.0_hk*z1**2
                    b = b*f
                else
                    b = 3.0_hk*f
                end if
            end if
        case default
            z1 = z - 1.0_hk
            if (abs(z1) < 1.0_hk) then
                b = 3.0_hk*z
                if (z1 > 0.0_hk) b = b - 6.0_hk*z1
                b = b*f*f
            end if
        end select
    end function host_basis

    !> The fit.  Arguments are validated by the caller (101 .. 106 as the reference, :716-781); returns
    !! ierror 0, or 107 when there are fewer rows than coefficients (suprls 33, :1650-1654), a pivot of the
    !! normal equations is not positive (the analogue of suprls 34) or the refinement does not contract.
    !! `hist(1:ncol)` receives the sparse-area histogram when `want_hist` (what the reference leaves in
    !! work(1:ncol), :879-907).  info: 1 data rows, 2 constraint rows, 3 refinement steps, 4 last relative
    !! correction, 5 smallest pivot, 9 residual norm (suprls :1693), 10 componentwise backward error.
    subroutine host_fit(ndim,xdata,l1xdat,ydata,wdata,weighted,ndata,xmin,xmax,nodes,xtrap,coef,hist,want_hist,info,ierror)
        integer,intent(in) :: ndim, l1xdat, ndata
        real(wp),intent(in) :: xdata(l1xdat,*), ydata(*), wdata(*), xmin(*), xmax(*), xtrap
        logical,intent(in) :: weighted, want_hist
        integer,intent(in) :: nodes(*)
        real(wp),intent(out) :: coef(*)
        real(wp),intent(inout) :: hist(*)
        real(real64),intent(out) :: info(10)
        integer,intent(out) :: ierror

        integer,parameter :: maxnz = 1024
        integer :: ncol, p, stride(ndim), inmx(ndim), nd(ndim)
        real(hk) :: dx(ndim), s(ndim), xlo(ndim)
        real(hk),allocatable :: ab(:,:), rhs(:), x(:), rho(:), den(:), hst(:)
        real(hk) :: ssq, totlwt, wtprrc, rel, prevrel, xmaxabs, dmax, pivmin, omega
        integer :: cols(maxnz), nz, idim, j, k, m, steps, i
        real(hk) :: vals(maxnz)
        integer(8) :: nrows_data, nrows_cons
        logical :: smooth

        ierror = 0
        info = 0.0_real64
        ncol = 1
        p = 0
        do idim = 1, ndim
            stride(idim) = ncol
            p = p + 3*ncol
            ncol = ncol*nodes(idim)
            nd(idim) = nodes(idim)
            inmx(idim) = nodes(idim) - 1
            xlo(idim) = real(xmin(idim),hk)
            dx(idim) = (real(xmax(idim),hk) - xlo(idim))/real(nodes(idim)-1,hk)      ! :747
            s(idim) = 1.0_hk/dx(idim)                                              ! :748
        end do
        p = min(p, ncol-1)
        smooth = xtrap /= 0.0_wp
        allocate(ab(0:p,ncol), rhs(ncol), x(ncol), rho(ncol), den(ncol), hst(ncol))
        ab = 0.0_hk
        rhs = 0.0_hk
        hst = 0.0_hk
        totlwt = 0.0_hk
        wtprrc = 0.0_hk
        if (smooth) call histogram()
        call rows(1)
        info(1) = real(nrows_data,real64)
        info(2) = real(nrows_cons,real64)
        if (want_hist) hist(1:ncol) = real(hst,wp)
        coef(1:ncol) = 0.0_wp
        if (nrows_data + nrows_cons < int(ncol,8)) then        ! suprls 33 -> 107
            ierror = 107
            return
        end if
        ! ---- band Cholesky, right-looking; column j of L lives in ab(0:m, j)
        pivmin = huge(1.0_hk)
        do j = 1, ncol
            if (.not. (ab(0,j) > 0.0_hk)) then
                ierror = 107
                return
            end if
            pivmin = min(pivmin, ab(0,j))
            ab(0,j) = sqrt(ab(0,j))
            m = min(p, ncol-j)
            ab(1:m,j) = ab(1:m,j)/ab(0,j)
            do k = 1, m
                if (ab(k,j) /= 0.0_hk) ab(0:m-k,j+k) = ab(0:m-k,j+k) - ab(k,j)*ab(k:m,j)
            end do
        end do
        info(5) = real(pivmin,real64)
        ! ---- solve + iterative refinement with the residual recomputed from the rows
        x = rhs
        call solve(x)
        steps = 0
        prevrel = huge(1.0_hk)
        rel = 0.0_hk
        do i = 1, 30
            call rows(2)                       ! rho = A^T (b - A x)
            call solve(rho)
            dmax = maxval(abs(rho))
            x = x + rho
            xmaxabs = maxval(abs(x))
            steps = steps + 1
            rel = 0.0_hk
            if (xmaxabs > 0.0_hk) rel = dmax/xmaxabs
            if (rel /= rel) exit
            if (rel <= 1.0e-13_hk) exit
            if (i >= 2 .and. rel >= 0.5_hk*prevrel) exit       ! at the rounding floor, or not contracting
            prevrel = rel
        end do
        info(3) = steps
        info(4) = real(rel,real64)
        if (rel /= rel .or. rel > 1.0e-8_hk) then
            ierror = 107
            return
        end if
        call rows(3)                           ! final pass: residual norm and backward error at the returned x
        omega = 0.0_hk
        do j = 1, ncol
            if (den(j) > 0.0_hk) omega = max(omega, abs(rho(j))/den(j))
        end do
        info(9) = real(sqrt(ssq),real64)
        info(10) = real(omega,real64)
        coef(1:ncol) = real(x,wp)

    contains

        !> x <- (L L^T)^{-1} x
        subroutine solve(v)
            real(hk),intent(inout) :: v(:)
            integer :: jj, mm
            do jj = 1, ncol
                mm = min(p, ncol-jj)
                v(jj) = v(jj)/ab(0,jj)
                if (v(jj) /= 0.0_hk) v(jj+1:jj+mm) = v(jj+1:jj+mm) - ab(1:mm,jj)*v(jj)
            end do
            do jj = ncol, 1, -1
                mm = min(p, ncol-jj)
                v(jj) = (v(jj) - dot_product(ab(1:mm,jj), v(jj+1:jj+mm)))/ab(0,jj)
            end do
        end subroutine solve

        !> nearest-node histogram of the weights with the reference's out-of-range quirk (:886-907: a dimension whose
        !! nearest node is outside the grid is SKIPPED in the Horner address, the point is still counted)
        subroutine histogram()
            integer :: ip, d, inidim, iin
            real(hk) :: bump, nrect
            do ip = 1, ndata
                bump = 1.0_hk
                if (weighted) bump = real(wdata(ip),hk)
                if (bump == 0.0_hk) cycle
                iin = 0
                do d = ndim, 1, -1
                    inidim = int(s(d)*(real(xdata(d,ip),hk) - xlo(d)) + 0.5_hk)
                    if (inidim < 0 .or. inidim > inmx(d)) cycle
                    iin = (inmx(d)+1)*iin + inidim
                end do
                hst(iin+1) = hst(iin+1) + bump
                totlwt = totlwt + bump
            end do
            nrect = 1.0_hk
            do d = 1, ndim
                nrect = nrect*real(inmx(d),hk)
            end do
            wtprrc = totlwt/nrect                                                  ! :910
        end subroutine histogram

        !> every row of the least-squares system, in the reference's order.  mode 1: N += a a^T, rhs += a b (assembly);
        !! mode 2: rho = A^T (b - A x); mode 3: as 2, plus the sum of squared residuals and the backward-error denominators
        subroutine rows(mode)
            integer,intent(in) :: mode
            integer :: ip, d, it, ibmn(ndim), ibmx(ndim), ib(ndim), in(ndim), nder(ndim), idm, jdm, iin
            real(hk) :: xx(ndim), tab(4,ndim), w, v, expect, dcwght, rowwt
            logical :: boundary
            if (mode >= 2) rho = 0.0_hk
            if (mode == 3) then
                den = 0.0_hk
                ssq = 0.0_hk
            end if
            nrows_data = 0
            nrows_cons = 0
            do ip = 1, ndata                                                       ! data rows, :788-855
                w = 1.0_hk
                if (weighted) w = real(wdata(ip),hk)
                if (w == 0.0_hk) cycle                                             ! :799
                do d = 1, ndim
                    xx(d) = real(xdata(d,ip),hk)
                    it = int(max(min(s(d)*(xx(d) - xlo(d)), 2.0e9_hk), -2.0e9_hk))   ! :821, truncation toward zero
                    ibmn(d) = min(max(it-1,0), nd(d)-2)
                    ibmx(d) = max(min(it+2,nd(d)-1), 1)
                    do j = ibmn(d), ibmx(d)
                        tab(j-ibmn(d)+1,d) = host_basis(j, nd(d), 0, xx(d), xlo(d) + real(j,hk)*dx(d), s(d))
                    end do
                end do
                nz = 0
                ib = ibmn
                do
                    v = w
                    k = 1
                    do d = 1, ndim
                        v = v*tab(ib(d)-ibmn(d)+1,d)
                        k = k + ib(d)*stride(d)
                    end do
                    nz = nz + 1
                    cols(nz) = k
                    vals(nz) = v
                    d = 1
                    do while (d <= ndim)
                        ib(d) = ib(d) + 1
                        if (ib(d) <= ibmx(d)) exit
                        ib(d) = ibmn(d)
                        d = d + 1
                    end do
                    if (d > ndim) exit
                end do
                call emit(mode, w*real(ydata(ip),hk))
                nrows_data = nrows_data + 1
            end do
            if (.not. smooth) return
            in = 0                                                                 ! constraint rows, :921-1046
            iin = 0
            do
                iin = iin + 1
                expect = wtprrc
                do d = 1, ndim
                    if (in(d) == 0 .or. in(d) == inmx(d)) expect = 0.5_hk*expect
                end do
                if (hst(iin) < 0.75_hk*expect) then                                ! spcrit, :696, :936
                    dcwght = (expect - hst(iin))*real(xtrap,hk)
                    do d = 1, ndim
                        xx(d) = xlo(d) + real(in(d),hk)*dx(d)
                        ibmn(d) = max(in(d)-1, 0)
                        ibmx(d) = min(in(d)+1, inmx(d))
                    end do
                    do idm = 1, ndim
                        do jdm = idm, ndim
                            nder = 0
                            boundary = .true.
                            rowwt = 2.0_hk*dcwght
                            if (jdm == idm) then
                                rowwt = dcwght
                                nder(jdm) = 2
                                if (in(idm) /= 0 .and. in(idm) /= inmx(idm)) boundary = .false.
                            end if
                            if (boundary) then
                                nder(idm) = 1
                                nder(jdm) = 1
                            end if
                            nz = 0
                            ib = ibmn
                            do
                                v = rowwt
                                k = 1
                                do d = 1, ndim
                                    v = v*host_basis(ib(d), nd(d), nder(d), xx(d), xlo(d) + real(ib(d),hk)*dx(d), s(d))
                                    k = k + ib(d)*stride(d)
                                end do
                                nz = nz + 1
                                cols(nz) = k
                                vals(nz) = v
                                d = 1
                                do while (d <= ndim)
                                    ib(d) = ib(d) + 1
                                    if (ib(d) <= ibmx(d)) exit
                                    ib(d) = ibmn(d)
                                    d = d + 1
                                end do
                                if (d > ndim) exit
                            end do
                            call emit(mode, 0.0_hk)
                            nrows_cons = nrows_cons + 1
                        end do
                    end do
                end if
                d = 1
                do while (d <= ndim)
                    in(d) = in(d) + 1
                    if (in(d) <= inmx(d)) exit
                    in(d) = 0
                    d = d + 1
                end do
                if (d > ndim) exit
            end do
        end subroutine rows

        !> one row (cols ascending, vals, right-hand side b)
        subroutine emit(mode, b)
            integer,intent(in) :: mode
            real(hk),intent(in) :: b
            integer :: a, c
            real(hk) :: dot, adot, res
            if (mode == 1) then
                do c = 1, nz
                    if (vals(c) == 0.0_hk) cycle
                    do a = c, nz
                        ab(cols(a)-cols(c), cols(c)) = ab(cols(a)-cols(c), cols(c)) + vals(a)*vals(c)
                    end do
                    rhs(cols(c)) = rhs(cols(c)) + vals(c)*b
                end do
                return
            end if
            dot = 0.0_hk
            adot = 0.0_hk
            do c = 1, nz
                dot = dot + vals(c)*x(cols(c))
                adot = adot + abs(vals(c))*abs(x(cols(c)))
            end do
            res = b - dot
            do c = 1, nz
                rho(cols(c)) = rho(cols(c)) + vals(c)*res
            end do
            if (mode == 3) then
                do c = 1, nz
                    den(cols(c)) = den(cols(c)) + abs(vals(c))*(adot + abs(b))
                end do
                ssq = ssq + res*res
            end if
        end subroutine emit

    end subroutine host_fit

end module splpak_host_solver
