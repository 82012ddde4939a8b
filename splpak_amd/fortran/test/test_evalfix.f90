!> The scalar (host) `evaluate` of the drop-in module against the REFERENCE's own splde values in 3-D and 4-D
!! (src/splpak.F90:1089-1240): text fixtures tests/golden/eval_<case>.txt, written by oracle/gen_eval_fixture.py from
!! the golden vectors of the unmodified reference -- coefficients, queries (exact node and boundary locations, points
!! outside the grid, the seeded stream) and one value per query and nderiv pattern (incl. second derivatives).
!! Needs no GPU: the scalar evaluate is a host computation, and `destroy(ndim)` prepares the object the way the
!! reference allows (SURVEY appendix C).   usage: test_evalfix <fixture.txt> [...]
program test_evalfix
    use splpak_module, wp => splpak_wp
    implicit none
    integer :: nargs, ia, nbad
    character(len=1024) :: path

    nbad = 0
    nargs = command_argument_count()
    if (nargs < 1) error stop 'usage: test_evalfix fixture.txt ...'
    do ia = 1, nargs
        call get_command_argument(ia, path)
        call one(trim(path))
    end do
    if (nbad /= 0) error stop 'FAIL test_evalfix'
    write(*,'(A)') ' PASS test_evalfix'
contains
    subroutine one(file)
        character(len=*),intent(in) :: file
        integer :: u, ndim, ncol, npat, nq, ip, iq, ierror, nder(8), nodes(8), k
        real(wp) :: xmin(8), xmax(8), x(8), vref, v, vmax, worst, cmax, scale
        real(wp),allocatable :: coef(:), xs(:,:), vr(:), vb(:)
        type(splpak_type) :: s
        open(newunit=u, file=file, status='old', action='read')
        read(u,*) ndim
        read(u,*) nodes(1:ndim)
        read(u,*) xmin(1:ndim)
        read(u,*) xmax(1:ndim)
        read(u,*) ncol, npat, nq
        allocate(coef(ncol), xs(ndim,nq), vr(nq))
        do k = 1, ncol
            read(u,*) coef(k)
        end do
        call s%destroy(ndim)
        cmax = maxval(abs(coef))
        worst = 0.0_wp
        do ip = 1, npat
            read(u,*) nder(1:ndim)
            vmax = 0.0_wp
            do iq = 1, nq
                read(u,*) xs(1:ndim,iq), vr(iq)
                vmax = max(vmax, abs(vr(iq)))
            end do
            ! scale = the size of the terms that are summed, |coef| * prod dxin**nderiv (as tests/test_gpu_parity.py):
            ! derivatives that cancel are judged against what cancels
            scale = cmax
            do k = 1, ndim
                scale = scale * (real(nodes(k) - 1, wp)/(xmax(k) - xmin(k)))**nder(k)
            end do
            vmax = max(vmax, scale)
            if (ndim > 4) then
                ! more than four dimensions: the BATCH call runs on the module's host solver by itself (the HIP kernels are written
                ! for 1..4 dimensions; the reference takes any ndim, src/splpak.F90:1166-1172)
                allocate(vb(nq))
                call s%evaluate_many(ndim, nq, xs, ndim, nder(1:ndim), coef, xmin(1:ndim), xmax(1:ndim), nodes(1:ndim), vb, ierror)
                if (ierror /= 0) nbad = nbad + 1
                do iq = 1, nq
                    if (abs(vb(iq) - vr(iq)) > 1.0e-12_wp*vmax) nbad = nbad + 1
                end do
                deallocate(vb)
            end if
            do iq = 1, nq
                x(1:ndim) = xs(:,iq)
                vref = vr(iq)
                v = s%evaluate(ndim, x(1:ndim), nder(1:ndim), coef, xmin(1:ndim), xmax(1:ndim), nodes(1:ndim), ierror)
                if (ierror /= 0) then
                    nbad = nbad + 1
                    write(*,*) 'ierror ', ierror, ' pattern ', ip, ' query ', iq
                end if
                worst = max(worst, abs(v - vref)/max(vmax, tiny(1.0_wp)))
                if (abs(v - vref) > 1.0e-12_wp*vmax) then
                    nbad = nbad + 1
                    if (nbad < 10) write(*,'(A,I3,A,I4,2ES25.16)') ' mismatch: pattern ', ip, ' query ', iq, v, vref
                end if
                if (all(nder(1:ndim) == 0)) then           ! splfe = splde with nderiv 0
                    v = s%evaluate(ndim, x(1:ndim), coef, xmin(1:ndim), xmax(1:ndim), nodes(1:ndim), ierror)
                    if (abs(v - vref) > 1.0e-12_wp*vmax) nbad = nbad + 1
                end if
            end do
        end do
        close(u)
        write(*,'(A,A,A,I3,A,I4,A,ES10.2)') ' ', file, ': ', npat, ' patterns x ', nq, ' queries, worst relative difference ', worst
    end subroutine one
end program test_evalfix
